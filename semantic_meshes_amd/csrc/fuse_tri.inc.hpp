// fuse_tri.inc.hpp -- device code of the triangle-order fusion kernel k_fuse_tri and the helpers it shares with the rest of
// fusion.hip.  Included INSIDE the anonymous namespace of fusion.hip (one view per launch, and everything else) and of
// fusion_pair.hip (the two-views-per-launch instances: a translation unit of their own so that the two halves compile in parallel).
// Not a stand-alone header.

constexpr int kWave = 64;

// ------------------------------------------------------------------------------------------------
// aggregator input maps (Fusion.cu:51-56 Summax, :70-73 Sum, :83-87 Mul)
// ------------------------------------------------------------------------------------------------
// Mul: LogProb<float>(pow(p, w)) (Fusion.cu:83-87) = log(p^w).  Whether the reference rounds p^w to float before the log is
// decided inside the absent template-tensors; the spec here -- the same one formula in the oracle -- is
// w * log(p) in float32, with p^0 = 1 for every p (so w == 0 contributes nothing, also for p = 0), p = 0 -> -inf,
// p < 0 -> NaN.  No pow: nothing underflows, no platform-dependent denormals, an eighth of the instructions.
// The log itself is part of the spec: a fixed sequence of IEEE float32 operations (Cephes' logf polynomial, every multiply-add an
// explicit fma), written out identically in oracle/smesh_oracle.cpp, so that the HIP path and the oracle compute bit-identical
// terms -- two different library logf's disagree by an ulp per term, which adds up to 1e-4 .. 1e-3 over the thousands of pixels of
// a large primitive.  Within 2 ulp of the correctly rounded natural logarithm (tests/test_oracle.py).
__device__ __forceinline__ float log_spec(float x) {
  if (!(x > 0.0f)) return x == 0.0f ? -INFINITY : NAN;          // log(0) = -inf; negative or NaN -> NaN
  if (x == INFINITY) return INFINITY;
  int e_adj = 0;
  uint32_t ix = __float_as_uint(x);
  if (ix < 0x00800000u) { x = x * 8388608.0f; e_adj = -23; ix = __float_as_uint(x); }   // denormal: scaled by 2^23 (exact)
  int e = (int)(ix >> 23) - 127 + e_adj;
  uint32_t im = (ix & 0x007FFFFFu) | 0x3F800000u;               // mantissa in [1, 2)
  if ((ix & 0x007FFFFFu) > 0x003504F3u) { im -= 0x00800000u; e += 1; }   // above sqrt(2): halve it -> m in (sqrt(1/2), sqrt(2)]
  const float f = __uint_as_float(im) - 1.0f;                   // exact
  const float z = f * f;
  float y = 7.0376836292e-2f;
  y = __builtin_fmaf(y, f, -1.1514610310e-1f);
  y = __builtin_fmaf(y, f, 1.1676998740e-1f);
  y = __builtin_fmaf(y, f, -1.2420140846e-1f);
  y = __builtin_fmaf(y, f, 1.4249322787e-1f);
  y = __builtin_fmaf(y, f, -1.6668057665e-1f);
  y = __builtin_fmaf(y, f, 2.0000714765e-1f);
  y = __builtin_fmaf(y, f, -2.4999993993e-1f);
  y = __builtin_fmaf(y, f, 3.3333331174e-1f);
  y = (y * f) * z;
  const float fe = (float)e;
  y = __builtin_fmaf(-2.12194440e-4f, fe, y);
  y = __builtin_fmaf(-0.5f, z, y);
  return __builtin_fmaf(0.693359375f, fe, f + y);
}

__device__ __forceinline__ float log_of_power(float p, float w) {
  return w == 0.0f ? 0.0f : w * log_spec(p);
}

// Mul state.  Rows are log-domain sums whose common offset cancels in get() (logprob_normalize divides by the largest element,
// Fusion.h:97-104), and so does it in the cross-GPU sum.  float32 sums of log-probabilities (the reference's LogProb<float>)
// lose what matters -- the small DIFFERENCE between the leading classes of a row -- to the ulp of the large sums themselves, so
// here a Mul row is the unevaluated sum of two float32 planes, acc (hi) + acc_lo, and
//   * k_fuse_tri / fuse_box sum a view's contributions separately, from zero, in double, and fold them into (hi, lo) once per
//     view in double: hi + lo carries ~48 bits;
//   * a row is re-centred on its largest finite element whenever a view adds to it, so that hi stays a small number for the
//     classes that matter -- which also keeps the kernels that only know the hi plane (any class count, texels, the generic
//     scatter-add: their float32 additions stay valid, hi + lo is still the value) as accurate as float32 allows.
template <int CT, bool EXACT>
__device__ __forceinline__ float row_centre(const float (&r)[CT], int C) {
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < CT; c++) if (EXACT || c < C) if (r[c] > m && r[c] < INFINITY) m = r[c];
  return m > -INFINITY ? m : 0.0f;
}

template <int KIND>
__device__ __forceinline__ float contribution(float p, float w) {
  if (KIND == SMESH_AGG_MUL) return log_of_power(p, w);
  return p * w;
}

// Orders this wave's LDS writes before its later LDS reads (LDS executes a wave's instructions in order;
// this only stops the compiler from reordering them).  No s_barrier: waves never wait for each other here.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef float f4 __attribute__((ext_vector_type(4)));

// Cross-lane moves in registers (DPP) instead of ds_bpermute: a dependent chain of LDS round trips per pixel made
// the wide-row kernels latency-bound.  Lanes whose source is disabled or outside the row keep `old`.
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ uint32_t dpp_u(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138;

__device__ __forceinline__ float wave_sum(float v) {   // same value in every lane (read back from lane 63)
  v += dpp_f<kDppRowShr1>(0.0f, v);
  v += dpp_f<kDppRowShr2>(0.0f, v);
  v += dpp_f<kDppRowShr4>(0.0f, v);
  v += dpp_f<kDppRowShr8>(0.0f, v);
  v += dpp_f<kDppRowBcast15, 0xA>(0.0f, v);
  v += dpp_f<kDppRowBcast31, 0xC>(0.0f, v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ double wave_sum_d(double v) {   // the same reduction tree on doubles (two 32-bit DPP moves per step)
#define SMESH_DPP_D(CTRL, MASK)                                                                                     \
  {                                                                                                                 \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, MASK, 0xF, false);                       \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, MASK, 0xF, false);                       \
    v += __hiloint2double(hi, lo);                                                                                  \
  }
  SMESH_DPP_D(kDppRowShr1, 0xF) SMESH_DPP_D(kDppRowShr2, 0xF) SMESH_DPP_D(kDppRowShr4, 0xF) SMESH_DPP_D(kDppRowShr8, 0xF)
  SMESH_DPP_D(kDppRowBcast15, 0xA) SMESH_DPP_D(kDppRowBcast31, 0xC)
#undef SMESH_DPP_D
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// Folds a view's contribution `part` into the (hi, lo) pair of one Mul accumulator element, the row shifted by -m.
__device__ __forceinline__ void mul_fold(float& hi, float& lo, const float m, const double part) {
  const double t = (((double)hi - (double)m) + (double)lo) + part;
  hi = (float)t;
  lo = (hi > -INFINITY && hi < INFINITY) ? (float)(t - (double)hi) : 0.0f;   // (-inf / NaN: no remainder)
}

__device__ __forceinline__ float wave_max(float v) {   // same value in every lane; lanes that must not take part pass -inf
  v = fmaxf(v, dpp_f<kDppRowShr1>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kDppRowShr2>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kDppRowShr4>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kDppRowShr8>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kDppRowBcast15, 0xA>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kDppRowBcast31, 0xC>(-INFINITY, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ uint32_t wave_scan_incl_u(uint32_t v) {   // lane l: v[0] + .. + v[l]
  v += dpp_u<kDppRowShr1>(0u, v);
  v += dpp_u<kDppRowShr2>(0u, v);
  v += dpp_u<kDppRowShr4>(0u, v);
  v += dpp_u<kDppRowShr8>(0u, v);
  v += dpp_u<kDppRowBcast15, 0xA>(0u, v);
  v += dpp_u<kDppRowBcast31, 0xC>(0u, v);
  return v;
}

__device__ __forceinline__ uint32_t wave_sum_u(uint32_t v) {
  v += dpp_u<kDppRowShr1>(0u, v);
  v += dpp_u<kDppRowShr2>(0u, v);
  v += dpp_u<kDppRowShr4>(0u, v);
  v += dpp_u<kDppRowShr8>(0u, v);
  v += dpp_u<kDppRowBcast15, 0xA>(0u, v);
  v += dpp_u<kDppRowBcast31, 0xC>(0u, v);
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// ------------------------------------------------------------------------------------------------
// Triangle-order fusion (smesh_fuse_view with a triangle renderer): the rasteriser left, per triangle, the set
// of pixels it emitted (TriFrag).  Every accumulator row is then owned by exactly one lane: NO atomics, no
// histogram pass (the per-view count is the number of the triangle's fragments that won the depth test), no
// sort/merge machinery, and the accumulator is read-modify-written once, sequentially, 64 rows per wave.
// The arithmetic is the reference's, in its order (Mesh.h:94-106): for Sum/Summax the result is bit-identical
// to the float32 CPU oracle run single-threaded.
// ------------------------------------------------------------------------------------------------
// (TriFuseArgs: common.hpp)

// float4 at 4-byte alignment: rows are only float-aligned; gfx950 global memory takes dwordx4 at any dword address.
// (A packed struct gets scalarised: its stores became one write request per lane and dword.)
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef fvec4 fvec4_a4 __attribute__((aligned(4)));

// k_fuse_tri comes in two flavours: EXACT (the class count is the template parameter: cfg 5 / 19 / 40) and run-time C <= CT
// (CT = 8, 16 .. 40: the register arrays are sized CT, loops are predicated with c < C).
template <int CT, bool EXACT>
__device__ __forceinline__ void load_row(const float* __restrict__ pr, int C, float (&p)[CT]) {
  if (EXACT) {
#pragma unroll
    for (int c = 0; c < CT; c++) p[c] = pr[c];   // (nontemporal loads here: 13 800 -> 8 400 views/s at cfg2, round 3)
  } else {
#pragma unroll
    for (int c = 0; c < CT; c += 4) {
      if (c + 4 <= C) {
        const fvec4 q = *reinterpret_cast<const fvec4_a4*>(pr + c);
        p[c] = q.x; p[c + 1] = q.y; p[c + 2] = q.z; p[c + 3] = q.w;
      } else {
#pragma unroll
        for (int t = 0; t < 4; t++) if (c + t < CT) p[c + t] = (c + t < C) ? pr[c + t] : 0.0f;
      }
    }
  }
}

// One triangle of one view, its pixels found by scanning the box [x0, x1] x [y0, y1] of the index image: one WAVE, lanes
// over the box; per-lane partial sums are combined by a butterfly over the wave and lane c owns class c of the row.
// `by_mask`: the box is the (at most) 8 x 8 box of a view in which the triangle is SMALL, and `mask` (bit dx * 8 + dy) already names its
// visible pixels there -- the record's mask after the tile resolve: the same pixels the scan would find, without the view's index plane
// (which a view without queued triangles of its own then need not write at all: raster.hip, RasterArgs::idx_optional == 2).
template <int CT, int KIND, bool EXACT>
__device__ __forceinline__ void fuse_box(const TriFuseArgs& a, const uint32_t f, const int x0, const int y0, const int x1, const int y1,
                                         uint32_t* __restrict__ lds_list, const bool by_mask = false, const unsigned long long mask = 0ull) {
  const int C = EXACT ? CT : (int)a.C;   // run-time class count, C <= CT
  const int l = threadIdx.x;
  const int bh = y1 - y0 + 1;
  const long long npx = (long long)(x1 - x0 + 1) * bh;
  // U pixels per lane and step: their index loads, then their class vectors, are in flight together
  constexpr int U = CT <= 24 ? 4 : 2;
  auto pix_of = [&](long long i) -> uint64_t { return (uint64_t)(x0 + (int)(i / bh)) * a.H + (uint64_t)(y0 + (int)(i % bh)); };
  // where pixel `pix` (= x * H + y) has its class vector: dense images at pix * C, strided ones (a network's (H,W,C) output seen
  // as (W,H,C)) at x * ps0 + y * ps1
  const bool dense_probs = a.ps1 == (uint32_t)C && a.ps0 == a.H * (uint32_t)C;
  auto probs_at = [&](uint64_t pix) -> const float* {
    if (dense_probs) return a.probs + pix * C;
    const uint32_t x = (uint32_t)pix / a.H, y = (uint32_t)pix - x * a.H;
    return a.probs + ((uint64_t)x * a.ps0 + (uint64_t)y * a.ps1);
  };
  // this wave owns the row: its current value is requested now and written back at the end (plain read-modify-write) -- unless the
  // launch also fuses medium views (a.mid: fuse_mid_entries adds to the rows of such triangles with float atomics at the same time),
  // then this wave adds its totals with atomics as well
  const bool shared_row = KIND != SMESH_AGG_MUL && a.mid != 0;
  float row_value = (l < C && !shared_row) ? a.acc[(uint64_t)f * C + l] : 0.0f;
  const bool one_step = npx <= (long long)kWave * U;   // the whole box in one round of index loads
  uint32_t mine_n = 0, hits = 0;                        // this lane's pixels of the triangle (bit u of `hits`: slot u of the one round)
  for (long long base = 0; base < npx; base += (long long)kWave * U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long long i = base + (long long)u * kWave + l;
      if (by_mask) {      // (wave-uniform; such a box is one round: one_step, n <= 64)
        v[u] = (i < npx && ((mask >> ((int)(i / bh) * 8 + (int)(i % bh))) & 1ull)) ? f : ~f;
      } else {
        v[u] = a.idx[i < npx ? pix_of(i) : pix_of(0)];
        if (!(i < npx)) v[u] = ~f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) if (v[u] == f) { mine_n++; hits |= 1u << u; }
  }
  const uint32_t upto = wave_scan_incl_u(mine_n);
  const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
  if (n == 0) return;
  const float w0 = a.iew * (1.0f / (float)n) + (1 - a.iew) * 1.0f;
  float centre = 0.0f;   // Mul: the row is re-centred on its largest finite element (see "Mul state"); lane c holds class c
  if (KIND == SMESH_AGG_MUL) {
    const float m = wave_max((l < C && row_value < INFINITY) ? row_value : -INFINITY);
    if (m > -INFINITY) centre = m;
  }
  typedef typename std::conditional<KIND == SMESH_AGG_MUL, double, float>::type bpart_t;   // Mul: per-lane partial sums in double
  bpart_t part[CT];
#pragma unroll
  for (int c = 0; c < CT; c++) part[c] = (bpart_t)0;
  auto write_back = [&](const bpart_t total) {   // lane l < C owns class l of the row
    if (KIND == SMESH_AGG_MUL) {
      float hi = row_value, lo = a.acc_lo[(uint64_t)f * C + l];
      mul_fold(hi, lo, centre, (double)total);
      a.acc[(uint64_t)f * C + l] = hi;
      a.acc_lo[(uint64_t)f * C + l] = lo;
    } else if (shared_row) {
      if ((float)total != 0.0f) unsafeAtomicAdd(&a.acc[(uint64_t)f * C + l], (float)total);
    } else {
      a.acc[(uint64_t)f * C + l] = row_value + (float)total;
    }
  };
  auto accumulate = [&](const float (&p)[CT], const bool hit, const float wt) {
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) if (EXACT || c < C) sum = sum + p[c];
    if (!(hit && sum > 0.5f)) return;
    const float w = w0 * wt;
    if (KIND == SMESH_AGG_SUMMAX) {
      float best = p[0];
      int am = 0;
#pragma unroll
      for (int c = 1; c < CT; c++) if (EXACT || c < C) if (p[c] > best) { best = p[c]; am = c; }
#pragma unroll
      for (int c = 0; c < CT; c++) if (EXACT || c < C) part[c] = (c == am) ? part[c] + p[c] * w : part[c];
    } else {
#pragma unroll
      for (int c = 0; c < CT; c++) if (EXACT || c < C) part[c] += (bpart_t)contribution<KIND>(p[c], w);
    }
  };
  if (one_step && n <= (uint32_t)kWave) {
    // Boxes of a few hundred pixels with at most 64 of them visible (the usual triangle just over the 8 x 8 limit): the hits are
    // compacted through LDS -- lane j takes the j-th visible pixel -- so every lane loads ONE class vector instead of U.
    uint32_t at = upto - mine_n;
#pragma unroll
    for (int u = 0; u < U; u++)
      if (hits & (1u << u)) lds_list[at++] = (uint32_t)pix_of((long long)u * kWave + l);   // pixel indices fit 32 bits (W, H <= 65536)
    wave_sync();
    const bool hit = (uint32_t)l < n;
    const uint64_t pix = lds_list[hit ? l : 0];
    wave_sync();   // the list is rewritten by this wave's next triangle
    float p[CT];
    load_row<CT, EXACT>(probs_at(pix), C, p);
    const float wt = (a.weights && hit) ? a.weights[pix] : 1.0f;
    accumulate(p, hit, wt);
    // part[] is now ONE pixel's contribution per lane: instead of a butterfly per class (6 DPP steps each), the n vectors go
    // through LDS as rows and lane c adds up column c, in the order of the compacted list
    float* rows = reinterpret_cast<float*>(lds_list);
    if (hit) {
#pragma unroll
      for (int c = 0; c < CT; c++) if (EXACT || c < C) rows[l * C + c] = (float)part[c];   // (one pixel's contribution: a float32 term)
    }
    wave_sync();
    bpart_t col = (bpart_t)0;
    if (l < C)
      for (uint32_t j = 0; j < n; j++) col += (bpart_t)rows[j * (uint32_t)C + (uint32_t)l];
    wave_sync();   // the rows are rewritten by this wave's next triangle
    if (l < C) write_back(col);
    return;
  } else {
    for (long long base = 0; base < npx; base += (long long)kWave * U) {
      uint64_t pix[U];
      bool hit[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const long long i = base + (long long)u * kWave + l;
        pix[u] = i < npx ? pix_of(i) : pix_of(0);
        hit[u] = a.idx[pix[u]] == f && i < npx;
      }
      float p[U][CT];
      float wt[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const float* __restrict__ pr = probs_at(hit[u] ? pix[u] : pix_of(0));   // unconditional: the loads overlap
        load_row<CT, EXACT>(pr, C, p[u]);
        wt[u] = (a.weights && hit[u]) ? a.weights[pix[u]] : 1.0f;
      }
#pragma unroll
      for (int u = 0; u < U; u++) accumulate(p[u], hit[u], wt[u]);
    }
  }
  bpart_t mine = (bpart_t)0;
#pragma unroll
  for (int c = 0; c < CT; c++) if (EXACT || c < C) {
    bpart_t v;
    if constexpr (KIND == SMESH_AGG_MUL) v = wave_sum_d(part[c]); else v = wave_sum(part[c]);
    if (l == c) mine = v;
  }
  if (l < C) write_back(mine);
}

// Triangles with a bounding box larger than 8 x 8 pixels: one WAVE per queued triangle (so, unlike the small-triangle
// path, the summation order is a tree).  Runs in the tail blocks of k_fuse_tri.  With several views in one launch a triangle
// that is big in ANY of them is left to one tail wave for ALL of them (first view first, as separate calls would do it), so that
// no other wave touches its row; the views in which it is small are scanned as 8 x 8 boxes.  A triangle queued by several views
// is taken from the queue of the first of them only.
__device__ __forceinline__ TriFuseArgs with_view(const TriFuseArgs& a, const TriView& w) {
  TriFuseArgs x = a;
  x.frags = w.frags; x.idx = w.idx; x.probs = w.probs; x.weights = w.weights; x.big_queue = w.big_queue; x.big_len = w.big_len;
  x.W = w.W; x.H = w.H; x.ps0 = w.ps0; x.ps1 = w.ps1;
  return x;
}

// Medium triangles -- a box of at most kMidBox pixels in every view of the launch -- are fused by fuse_mid_entries (fuse_mid.inc.hpp: sixteen
// lanes per triangle, four triangles per wave) when TriFuseArgs::mid is set; the tail waves below then leave them alone.
__device__ __forceinline__ bool mid_box(const TriFrag& rec, uint32_t W, uint32_t H) {
  (void)W; (void)H;
  if (rec.kind != 2) return true;                        // nothing emitted here, or a box of at most 8 x 8
  const int w = (int)(rec.mask & 0xFFFFu) - (int)rec.x0 + 1, h = (int)((rec.mask >> 16) & 0xFFFFu) - (int)rec.y0 + 1;
  return w * h <= kMidBox;
}

template <int CT, int KIND, bool EXACT, int NV>
__device__ __forceinline__ void fuse_big_triangles(const TriFuseArgs& a, const TriViews<NV>& vw, uint32_t worker, uint32_t nworkers,
                                                   uint32_t* __restrict__ lds_list) {
  uint32_t len[NV], total = 0u;
#pragma unroll
  for (int v = 0; v < NV; v++) { len[v] = min(*vw.v[v].big_len, a.big_capacity); total += len[v]; }
  const int l = threadIdx.x;
  // `chunk` queue entries per step, one per lane: which of them are this wave's to fuse is decided for all of them at once (the
  // entry, its record, the records of the earlier views: loads in flight together), then the wave takes the chosen ones one after
  // the other.  (Round 2 walked the entries one per step: two dependent round trips per entry, seven out of eight of them
  // rejected -- a triangle sits in the queue of every view in which it is big and is taken from the first.)  The chunk grows
  // with the queues -- one entry per step while there are fewer entries than tail waves, so that a few expensive triangles
  // still spread over all of them, 64 when there are many.
  // With a.mid the MEDIUM views (boxes of at most kMidBox pixels) are fuse_mid_entries', whose waves run in the same launch and add with
  // float atomics: an entry counts only if its triangle is LARGE (over kMidBox pixels) in the entry's view, and is taken from the first
  // view in which it is large; the wave then adds with atomics too (fuse_box, a.mid).
  // A step's `chunk` entries are spread over the whole concatenation of the queues (lane l takes entry l * steps + step): the entries of
  // the first view's queue are all taken, the later views' mostly are duplicates, so consecutive entries would hand a few waves all
  // the work.
  const uint32_t chunk = max(1u, min((uint32_t)kWave, total / max(nworkers, 1u)));
  const uint32_t steps = (total + chunk - 1u) / chunk;
  for (uint32_t step = worker; step < steps; step += nworkers) {
    const uint32_t q = (uint32_t)l < chunk ? (uint32_t)l * steps + step : total;
    uint32_t fi = 0u;
    bool take = false;
    {
      uint32_t qq = q;
      bool located = q >= total;
      int jsel = -1;
#pragma unroll
      for (int j = 0; j < NV; j++) {
        if (!located) {
          if (qq < len[j]) { fi = vw.v[j].big_queue[qq]; jsel = j; located = true; }
          else qq -= len[j];
        }
      }
      if (jsel >= 0) {
        bool mine = false, earlier = false;
#pragma unroll
        for (int i = 0; i < NV; i++) {
          if (i <= jsel) {
            const TriFrag rec = vw.v[i].frags[fi];
            const bool counts = rec.kind == 2 && !(a.mid && mid_box(rec, vw.v[i].W, vw.v[i].H));
            if (i < jsel) earlier = earlier || counts;
            else mine = counts;
          }
        }
        take = mine && !earlier && fi >= a.f_lo && fi < a.f_hi;   // (f_lo, f_hi: fusion by triangle range; 0, F otherwise)
      }
    }
    unsigned long long todo = __ballot(take);
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      todo &= todo - 1ull;
      const uint32_t fq = (uint32_t)__builtin_amdgcn_readlane((int)fi, src);   // wave-uniform
      const uint32_t f = a.prim_id ? a.prim_id[fq] : fq;        // primitive id = value in the index image = accumulator row
#pragma unroll
      for (int j = 0; j < NV; j++) {
        const TriFrag rec = vw.v[j].frags[fq];
        if (rec.kind == 0 || (a.mid && rec.kind == 2 && mid_box(rec, 0u, 0u))) continue;   // (a medium view: fuse_mid_entries')
        const TriFuseArgs x = with_view(a, vw.v[j]);
        int x1, y1;
        if (rec.kind == 2) { x1 = (int)(rec.mask & 0xFFFFu); y1 = (int)((rec.mask >> 16) & 0xFFFFu); }
        else { x1 = min((int)rec.x0 + 7, (int)x.W - 1); y1 = min((int)rec.y0 + 7, (int)x.H - 1); }
        // (a view in which the triangle is small: its record's mask names the visible pixels, unless the render says its masks need checking)
        const bool by_mask = rec.kind == 1 && !a.prim_id && vw.v[j].big_len[1] == 0u;
        fuse_box<CT, KIND, EXACT>(x, f, rec.x0, rec.y0, x1, y1, lds_list, by_mask, rec.mask);
      }
    }
  }
}

#include "fuse_mid.inc.hpp"

// NV views (1, 2, 4 or 8) of the same mesh into the same accumulator in ONE launch, in order: the wave's 64-row block makes one
// round trip for all of them (the accumulator traffic is 15 us of a two-view launch's 83 at cfg2), and the additions happen in the
// order NV launches would have made them.  `a`: what the views share (accumulator, mesh, class count ...); `vw`: the views.
// (Register budget: the instruction scheduler trades registers for latency hiding in steps of whole waves per SIMD, and the eight-view
// Summax instance for 19 classes sits at 167 of the 168 registers that three waves allow -- any addition to the kernel, however far
// from its hot loop, dropped it to two waves at 183-185.  The budget of the instances for 19 .. 21 classes is therefore stated
// (20 and 21 classes / Summax gain their third wave that way: 169 / 175 registers without it); every other instance is left to the scheduler.  A budget of FOUR waves for the headline instance
// (19 classes, Sum, eight views: 128 registers, 64 spilled) measured slower: 41.3 vs 37.6 us per view.)
constexpr int fuse_tri_min_waves(int ct, int kind, int nv) { return (ct >= 19 && ct <= 21 && kind != SMESH_AGG_MUL && (nv != 2 || ct == 19)) ? 3 : 1; }   // (two views, 20 / 21 classes, Summax: would spill)
template <int CT, int KIND, bool EXACT, int NV>
__global__ __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(fuse_tri_min_waves(CT, KIND, NV)))) void k_fuse_tri(TriFuseArgs a, TriViews<NV> vw) {
  const int C = EXACT ? CT : (int)a.C;   // run-time class count, C <= CT
  constexpr int PB = CT <= 24 ? 2 : 1;        // pixels whose class vectors are in flight together (3 or 4: no difference, round 2)
  constexpr int KV = (kWave * CT / 4 + kWave - 1) / kWave;   // float4 per lane of the 64-row block
  constexpr int kSrow = (kWave * CT + 4) > (4 * kMidHits / 2) ? (kWave * CT + 4) : (4 * kMidHits / 2);   // (the medium-triangle waves: 4 x kMidHits uint16)
  __shared__ __attribute__((aligned(16))) float srow[kSrow];   // the wave's 64 accumulator rows
  const int l = threadIdx.x;
  // The launch's workgroups: a.tri_blocks main waves, one per 64 triangles; then a.big_blocks waves on the queued big triangles; then
  // (a.mid) the waves of the medium triangles, gone at once where the lists are empty.  (The medium-triangle waves FIRST, so that they
  // start beside the main waves: measured, no faster -- 90 000 triangles 0.105 vs 0.108 ms per view -- and 20 bytes of scratch.)
  if (blockIdx.x >= a.tri_blocks) {
    const uint32_t tail = blockIdx.x - a.tri_blocks;
    if (tail < a.big_blocks) {
      uint32_t* lds_list = reinterpret_cast<uint32_t*>(srow);   // the tail waves have no use for the row block: >= 64 entries
      fuse_big_triangles<CT, KIND, EXACT, NV>(a, vw, tail, a.big_blocks, lds_list);
    } else if constexpr (KIND != SMESH_AGG_MUL) {
      constexpr int MCT = (CT + 7) / 8 * 8;                     // class-vector register slots of the medium-triangle code: 8, 16 .. 48
      fuse_mid_entries<MCT, KIND, NV>(a, vw, tail - a.big_blocks, gridDim.x - a.tri_blocks - a.big_blocks, (lds_u16*)srow);
    }
    return;
  }
  const uint64_t f0 = ((uint64_t)a.blk_first + blockIdx.x) * kWave;   // (blk_first: fusion by triangle range; 0 otherwise)
  const uint64_t f = f0 + l;
  // per view: box origin (x0 | y0 << 16) and the mask of this triangle's VISIBLE pixels inside its <= 8 x 8 box
  uint32_t org[NV];
  unsigned long long msk[NV];
  // `big`: a box over 8 x 8 in some view -- other waves of this launch add to the row (the tail waves of the big triangles, the
  // medium-triangle waves with float atomics), so it is never stored from here.  `large`: a view for the tail waves (every such view
  // without a.mid; with it, the ones over kMidBox pixels): the tail wave then takes the triangle's small views as well.  A triangle
  // that is only MEDIUM somewhere keeps its small views in this lane, whose share goes to the row by float atomics at the end.
  bool big = false, large = false;
#pragma unroll
  for (int v = 0; v < NV; v++) { org[v] = 0u; msk[v] = 0ull; }
  if (f < a.F) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const TriFrag rec = vw.v[v].frags[f];
      org[v] = (uint32_t)rec.x0 | ((uint32_t)rec.y0 << 16);
      msk[v] = rec.kind == 1 ? rec.mask : 0ull;
      big = big || rec.kind == 2;
      large = large || (rec.kind == 2 && !(KIND != SMESH_AGG_MUL && a.mid && mid_box(rec, 0u, 0u)));
    }
  }
  if (large) {
#pragma unroll
    for (int v = 0; v < NV; v++) msk[v] = 0ull;
  }
  // Re-ordered mesh (renderer's position -> primitive id table): the id is what the index image holds and which row to
  // update; the 64 rows of a wave are then scattered, so every lane loads / stores its own row instead of the LDS block.
  const bool scattered = a.prim_id != nullptr;
  const uint32_t pid = (scattered && f < a.F) ? a.prim_id[f] : (uint32_t)f;
  auto pixel = [&](int v, int k) -> uint64_t {
    return (uint64_t)((org[v] & 0xFFFFu) + (uint32_t)(k >> 3)) * vw.v[v].H + (org[v] >> 16) + (uint32_t)(k & 7);
  };

  // ---- pass 1, normally skipped: the tile resolve of the rasteriser has already cleared the losers of the depth test out of the
  // masks (raster.hip, tile_resolve_block) unless the render says otherwise (big_len[1]: fragment-queue overflow, direct
  // rasteriser) or the mesh was re-ordered: then the candidates are checked against the index plane here, four index loads in
  // flight per lane.
  bool verify = scattered;
#pragma unroll
  for (int v = 0; v < NV; v++) verify = verify || vw.v[v].big_len[1] != 0u;
  if (verify) {
#pragma unroll
    for (int v = 0; v < NV; v++) {
      // (only the views whose OWN masks need checking: a view without the flag may not even have written its plane -- raster.hip,
      // RasterArgs::idx_optional; round 6's sweep caught the all-views loop on a scene where one view of the launch overflowed its queues)
      if (!scattered && vw.v[v].big_len[1] == 0u) continue;
      const uint32_t* __restrict__ idx = vw.v[v].idx;
      unsigned long long m = msk[v], win = 0ull;
      while (__ballot(m != 0ull) != 0ull) {
        int k[4];
        uint32_t got[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          k[j] = -1;
          if (m) { k[j] = __ffsll((long long)m) - 1; m &= m - 1ull; }
          got[j] = idx[k[j] >= 0 ? pixel(v, k[j]) : 0];   // unconditional (clamped) so that the loads overlap
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
          if (k[j] >= 0 && got[j] == pid) win |= 1ull << k[j];
      }
      msk[v] = win;
    }
  }
  unsigned long long any_win = 0ull;
#pragma unroll
  for (int v = 0; v < NV; v++) any_win |= msk[v];
  if (__ballot(any_win != 0ull) == 0ull) return;   // nothing of these 64 triangles is visible: rows untouched

  // ---- issue together: the wave's 64 accumulator rows (one contiguous block) and the first PB pixels'
  // class vectors of every lane
  const int nrows = (int)min((uint64_t)kWave, a.F - f0);
  float* __restrict__ blk = a.acc + f0 * C;
  f4 br[KV];
  if (nrows == kWave && !scattered) {
    const f4* b4 = reinterpret_cast<const f4*>(blk);
#pragma unroll
    for (int q = 0; q < KV; q++) br[q] = b4[min(l + q * kWave, kWave * C / 4 - 1)];
  }
  float accr[CT];
  bool rows_loaded = false;
  // Mul: the view's contributions are summed from zero in double and meet the re-centred row once.  (Until round 5 in float32 beyond
  // 24 register slots: the Mul instances for 25 .. 48 classes sit at 270 - 330 registers, one wave per SIMD, with or without the
  // 40 further ones -- and add_many's test found an element at 1.4e-5 of the float64 oracle at 40 classes.)
  typedef double part_t;
  constexpr int PT = KIND == SMESH_AGG_MUL ? CT : 1;
#pragma unroll
  for (int v = 0; v < NV; v++) {
    part_t part[PT];
#pragma unroll
    for (int c = 0; c < PT; c++) part[c] = (part_t)0;
    const float* __restrict__ probs = vw.v[v].probs;
    const float* __restrict__ weights = vw.v[v].weights;
    const uint32_t nv = (uint32_t)__popcll(msk[v]);   // this primitive's pixels in this view: the histogram entry of Mesh.h:90-93
    float w0 = 0.0f;
    if (nv) {
      const float image_weight = 1.0f / ((float)nv);                         // Mesh.h:100
      const float pixel_w = 1.0f;                                            // :101
      w0 = a.iew * image_weight + (1 - a.iew) * pixel_w;                     // :102
    }
    unsigned long long mm = msk[v];
    while (__ballot(mm != 0ull) != 0ull) {
      float p[PB][CT];
      float wt[PB];
      bool have[PB];
#pragma unroll
      for (int j = 0; j < PB; j++) {
        have[j] = mm != 0ull;
        int k = 0;
        if (mm) { k = __ffsll((long long)mm) - 1; mm &= mm - 1ull; }
        const uint64_t pix = have[j] ? pixel(v, k) : 0;
        // (dense images: ps0 = H * C, ps1 = C; the strided (H,W,C) output of a network is read in place, colorize_cityscapes_mesh.py:65-67)
        const float* __restrict__ pr = probs + (have[j] ? (uint64_t)((org[v] & 0xFFFFu) + (uint32_t)(k >> 3)) * vw.v[v].ps0 +
                                                          (uint64_t)((org[v] >> 16) + (uint32_t)(k & 7)) * vw.v[v].ps1 : 0);
        load_row<CT, EXACT>(pr, C, p[j]);
        wt[j] = weights ? weights[pix] : 1.0f;
      }
      if (!rows_loaded && scattered) {
        if (any_win) load_row<CT, EXACT>(a.acc + (uint64_t)pid * C, C, accr);
        if (KIND != SMESH_AGG_MUL && big) {   // (a row that takes atomics meanwhile: this lane sums its share from zero, see the end)
#pragma unroll
          for (int c = 0; c < CT; c++) accr[c] = 0.0f;
        }
        rows_loaded = true;
      }
      if (!rows_loaded) {
        // park the block in LDS (flat, coalesced) and pick up this lane's row
        if (nrows == kWave) {
          f4* s4 = reinterpret_cast<f4*>(srow);
#pragma unroll
          for (int q = 0; q < KV; q++)
            if (l + q * kWave < kWave * C / 4) s4[l + q * kWave] = br[q];
        } else {
          for (int q = l; q < nrows * C; q += kWave) srow[q] = blk[q];
        }
        wave_sync();
        if (KIND != SMESH_AGG_MUL && big) {   // (a row that takes atomics meanwhile: this lane sums its share from zero, see the end;
                                              // zeroed in LDS rather than by a select per register, which cost the Summax instances a wave per SIMD)
          for (int c = 0; c < C; c++) srow[l * C + c] = 0.0f;
        }
#pragma unroll
        for (int c = 0; c < CT; c++) if (EXACT || c < C) accr[c] = srow[l * C + c];
        rows_loaded = true;
      }
      // Mesh.h:94-106 for this primitive's pixels, in image order (x, then y)
#pragma unroll
      for (int j = 0; j < PB; j++) {
        float sum = 0.0f;
#pragma unroll
        for (int c = 0; c < CT; c++) if (EXACT || c < C) sum = sum + p[j][c];                      // tt::sum, sequential float32
        if (have[j] && sum > 0.5f) {                                          // :98
          const float w = w0 * wt[j];                                         // :103
          if (KIND == SMESH_AGG_SUMMAX) {
            int am = 0;
            float best = p[j][0];   // (not p[j][am]: a run-time register index would go through scratch)
#pragma unroll
            for (int c = 1; c < CT; c++) if (EXACT || c < C) if (p[j][c] > best) { best = p[j][c]; am = c; }
#pragma unroll
            for (int c = 0; c < CT; c++) if (EXACT || c < C) if (c == am) accr[c] = accr[c] + p[j][c] * w;
          } else if (KIND == SMESH_AGG_MUL) {
#pragma unroll
            for (int c = 0; c < PT; c++) if (EXACT || c < C) part[c] = part[c] + (part_t)contribution<KIND>(p[j][c], w);
          } else {
#pragma unroll
            for (int c = 0; c < CT; c++) if (EXACT || c < C) accr[c] = accr[c] + contribution<KIND>(p[j][c], w);
          }
        }
      }
    }
    if (KIND == SMESH_AGG_MUL && nv) {   // (nv != 0: the loop above ran, so this lane's row is in accr)
      const float m = row_centre<CT, EXACT>(accr, C);
      float* __restrict__ lo_row = a.acc_lo + (uint64_t)pid * C;   // the lo plane is read and written by the row's owner lane
      float lo[PT];
#pragma unroll
      for (int c = 0; c < PT; c++) lo[c] = (EXACT || c < C) ? lo_row[c] : 0.0f;
#pragma unroll
      for (int c = 0; c < PT; c++) if (EXACT || c < C) mul_fold(accr[c], lo[c], m, (double)part[c]);
#pragma unroll
      for (int c = 0; c < PT; c++) if (EXACT || c < C) lo_row[c] = lo[c];
    }
  }
  if (scattered && __ballot(big && any_win) == 0ull) {
    if (any_win) {
#pragma unroll
      for (int c = 0; c < CT; c++) if (EXACT || c < C) a.acc[(uint64_t)pid * C + c] = accr[c];
    }
    return;
  }
  // every other way out goes through LDS: each lane parks its row ...
#pragma unroll
  for (int c = 0; c < CT; c++) if (EXACT || c < C) srow[l * C + c] = accr[c];
  wave_sync();
  if (scattered || __ballot(big) != 0ull) {
    // Some of these 64 rows belong to big or medium triangles, which the tail blocks of this launch update concurrently: writing the
    // whole block back would overwrite their sums.  Every other lane stores its own row; the lane of a triangle that is medium in some
    // views (the medium-triangle waves are adding to its row with float atomics) and small in others adds what its small views came
    // to -- summed from zero, see above -- with atomics too.  Rolled loops over the parked row: unrolled over the register array, the
    // atomics cost the Summax instances for 19 .. 21 classes their third wave per SIMD (167 -> 183 VGPRs).
    float* __restrict__ row = a.acc + (uint64_t)pid * C;
    const float* __restrict__ mine = srow + l * C;
    if (KIND != SMESH_AGG_MUL && big) {
      if (any_win)
        for (int c = 0; c < C; c++) { const float d = mine[c]; if (d != 0.0f) unsafeAtomicAdd(&row[c], d); }
    } else if (!big && f < a.F && (any_win || !scattered)) {
      for (int c = 0; c < C; c++) row[c] = mine[c];
    }
    return;
  }
  if (nrows == kWave) {
    f4* b4 = reinterpret_cast<f4*>(blk);
    const f4* s4 = reinterpret_cast<const f4*>(srow);
    for (int q = l; q < kWave * C / 4; q += kWave) b4[q] = s4[q];
  } else {
    for (int q = l; q < nrows * C; q += kWave) blk[q] = srow[q];
  }
}

