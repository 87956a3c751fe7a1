// fuse_mid.inc.hpp -- the triangle-order fusion of MEDIUM triangles (a bounding box over 8 x 8 pixels, of at most kMidBox pixels):
// sixteen lanes per (triangle, view), four of them per wave, float atomics on the accumulator row.  Included INSIDE the anonymous
// namespace of the translation units that instantiate k_fuse_tri, after fuse_tri.inc.hpp's helpers: since round 4 the medium
// triangles are fused by extra one-wave workgroups at the END OF THE k_fuse_tri LAUNCH (fuse_mid_entries) instead of a launch of
// their own ahead of it (k_fuse_mid, round 3) -- at cfg2, where no triangle is medium, that launch cost 5.6 us of kernel time plus a
// launch gap per group of eight views, and on meshes that do have medium triangles the two kernels now share
// the chip instead of running one after the other.
//
// Why sixteen lanes per entry (NOTES/round3.md): a mesh of 90 000 triangles at 1080p (~23 pixels per triangle, boxes
// just over 8 x 8 -- a decimated indoor scan, eval-scannet/simplify_scannet_meshes.py) spent 115 us per view in the tail blocks of
// k_fuse_tri, where ONE WAVE takes one queued triangle at a time through all the views of the launch (fuse_box): a chain of dependent
// memory round trips (queue entry -> records -> index plane -> class vectors -> row) with a quarter of the lanes busy -- and seven
// queue entries out of eight are looked up only to be dropped (a triangle sits in the queue of every view in which it is big).
// Here the unit of work is the queue ENTRY: one (triangle, view).  It gets one 16-lane DPP row: the box is scanned by sixteen lanes
// (all index loads of a lane in flight together), each lane adds up the weighted class vectors of its own hits, the row-wide sums
// are an all-reduce by row rotations (`row_ror` 8, 4, 2, 1) and lane l adds classes l, l + 16, l + 32 to the accumulator row with
// float atomics -- the views of a triangle meet in its row in any order, so no entry has to know about the others.
// Sum and Summax only (Mul's (hi, lo) rows cannot take atomics: those aggregators keep the one-wave-per-triangle tail).
// The arithmetic is Mesh.h:94-106 term for term; the ORDER of the additions is a tree per view and arbitrary across views:
// 1e-5 like every path that is not one-lane-per-row, and not run-to-run deterministic to the last bit.
//
// Work split (TriFuseArgs::mid != 0): a view in which the triangle's box holds at most kMidBox pixels (mid_box()) is fused here;
// k_fuse_tri's main waves leave every row alone whose triangle has a box over 8 x 8 in some view of the launch (these waves add to
// it at the same time); its tail waves take such a triangle's large views and its small views, and likewise add with atomics
// (fuse_box, a.mid).

constexpr int kMidHits = 256;     // compacted hits per 16-lane row = kMidBox: every box fits (LDS: 16 rows x 256 x 2 bytes per workgroup)
constexpr int kRowRor1 = 0x121, kRowRor2 = 0x122, kRowRor4 = 0x124, kRowRor8 = 0x128;

__device__ __forceinline__ float row16_sum(float v) {     // every lane of the 16-lane row ends with the row's total
  v += dpp_f<kRowRor8>(0.0f, v);
  v += dpp_f<kRowRor4>(0.0f, v);
  v += dpp_f<kRowRor2>(0.0f, v);
  v += dpp_f<kRowRor1>(0.0f, v);
  return v;
}
__device__ __forceinline__ uint32_t row16_sum_u(uint32_t v) {
  v += dpp_u<kRowRor8>(0u, v);
  v += dpp_u<kRowRor4>(0u, v);
  v += dpp_u<kRowRor2>(0u, v);
  v += dpp_u<kRowRor1>(0u, v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<kRowRor8>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kRowRor4>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kRowRor2>(-INFINITY, v));
  v = fmaxf(v, dpp_f<kRowRor1>(-INFINITY, v));
  return v;
}
__device__ __forceinline__ double row16_sum_d(double v) {
#define SMESH_ROR_D(CTRL)                                                                               \
  {                                                                                                     \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);            \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);            \
    v += __hiloint2double(hi, lo);                                                                      \
  }
  SMESH_ROR_D(kRowRor8) SMESH_ROR_D(kRowRor4) SMESH_ROR_D(kRowRor2) SMESH_ROR_D(kRowRor1)
#undef SMESH_ROR_D
  return v;
}

// CT: register slots of a class vector (C <= CT, run-time C).  One WAVE = four entries in flight; `worker` of `nworkers` such waves
// (the last workgroups of a k_fuse_tri launch).  `hit_lds`: 4 x kMidHits uint16 of this wave's LDS.
typedef __attribute__((address_space(3))) uint16_t lds_u16;
template <int CT, int KIND, int NV>
__device__ __forceinline__ void fuse_mid_entries(const TriFuseArgs& a, const TriViews<NV>& vw, const uint32_t worker, const uint32_t nworkers,
                                                 lds_u16* __restrict__ hit_lds) {
  static_assert(KIND != SMESH_AGG_MUL, "fuse_mid_entries adds with float atomics: Sum and Summax only");
  const int C = (int)a.C;
  const int l16 = (int)(threadIdx.x & 15u);
  const uint32_t slot = worker * 4u + (threadIdx.x >> 4), nslots = nworkers * 4u;
  uint32_t len[NV], total = 0u;
#pragma unroll
  for (int v = 0; v < NV; v++) { len[v] = min(vw.v[v].big_len[3], a.big_capacity); total += len[v]; }   // the rasteriser's lists of medium triangles (push_mid)
  if (total == 0u) return;             // no medium triangle in any view: every BASELINE config
  lds_u16* __restrict__ my_hits = hit_lds + ((threadIdx.x >> 4) & 3u) * kMidHits;   // per 16-lane row: the box pixels that hold the triangle, compacted (pass 2 takes one per lane and round)
  // what a queue entry is: the view it belongs to, the triangle, the triangle's record in that view
  struct Entry { uint32_t fi; int view; TriFrag rec; };
  auto fetch = [&](const uint32_t q) -> Entry {
    Entry e;
    e.fi = 0u; e.view = -1;
    e.rec.x0 = 0; e.rec.y0 = 0; e.rec.kind = 0; e.rec.pad = 0; e.rec.mask = 0ull;
    // which view's list entry q falls into: arithmetic on the eight lengths; the view's pointers are then read from the kernel-argument
    // segment at a run-time index (chains of selects over all views held every pointer of every view in scalar registers, and the
    // spilled ones cost the whole kernel -- main path included -- six vector registers: one wave per SIMD at C = 19 / Summax)
    uint32_t qq = q;
    int jsel = -1;
    bool located = q >= total;
#pragma unroll
    for (int j = 0; j < NV; j++) {
      if (!located) {
        if (qq < len[j]) { jsel = j; located = true; }
        else qq -= len[j];
      }
    }
    if (jsel >= 0) {
      const TriView& w = vw.v[jsel];
      e.fi = w.big_queue[(uint64_t)a.big_capacity + qq];
      e.rec = w.frags[e.fi];
      e.view = jsel;
    }
    return e;
  };
  Entry next = fetch(slot);
  for (uint32_t q0 = 0; q0 < total; q0 += nslots) {
    const uint32_t q = q0 + slot;
    // ---- this row's queue entry; the NEXT one is requested now, two round trips ahead of its use
    const Entry cur = next;
    next = fetch(q + nslots);
    const uint32_t fi = cur.fi;
    const TriFrag rec = cur.rec;
    const TriView& cw = vw.v[cur.view >= 0 ? cur.view : 0];
    const uint32_t* __restrict__ idx = cw.idx;
    const float* __restrict__ probs = cw.probs;
    const float* __restrict__ weights = cw.weights;
    const uint32_t vH = cw.H, ps0 = cw.ps0, ps1 = cw.ps1;
    const bool on = cur.view >= 0 && rec.kind == 2 && mid_box(rec, 0u, 0u);   // (larger boxes: k_fuse_tri's tail waves)
    if (__ballot(on) == 0ull) continue;
    const uint32_t pid = (a.prim_id && on) ? a.prim_id[fi] : fi;   // value in the index image = accumulator row
    const int x0 = rec.x0, y0 = rec.y0;
    const int x1 = (int)(rec.mask & 0xFFFFu), y1 = (int)((rec.mask >> 16) & 0xFFFFu);
    const int bh = on ? y1 - y0 + 1 : 1;
    const int npx = on ? (x1 - x0 + 1) * bh : 0;
    // pass 1: this lane's pixels l, l + 16, l + 32 ... of the box (x-major, y fastest) against the index plane -- all the loads
    // of a lane in flight together -- leave a 64-bit mask of its hits (a box holds at most kMidBox = 16 x 16 pixels: 16 per lane)
    unsigned long long hits = 0ull;
    {
      int cx = l16 / bh, cy = l16 - cx * bh;     // walks the lane's pixels without a division per pixel
      for (int base = 0; __ballot(base < npx) != 0ull; base += 16 * 8) {
        uint32_t got[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const bool in = base + u * 16 + l16 < npx;
          got[u] = idx[(uint64_t)(uint32_t)(x0 + (in ? cx : 0)) * vH + (uint32_t)(y0 + (in ? cy : 0))];
          if (!in) got[u] = ~pid;
          cy += 16;
          while (cy >= bh) { cy -= bh; cx++; }      // (bh >= 1; at most 16 steps, usually one or two)
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (got[u] == pid) hits |= 1ull << (base / 16 + u);
      }
    }
    const uint32_t mine_n = (uint32_t)__popcll(hits);
    const uint32_t ntot = row16_sum_u(mine_n);                        // the histogram entry of Mesh.h:90-93
    const float w0 = ntot ? a.iew * (1.0f / (float)ntot) + (1 - a.iew) * 1.0f : 0.0f;    // Mesh.h:100-102
    // The hits of the sixteen lanes, compacted through LDS (a DPP row scan gives every lane its offset): pass 2 then runs
    // ceil(ntot / 16) rounds with full rows instead of as many rounds as the unluckiest lane of the WAVE has hits.  Rows with more
    // than kMidHits hits (none while kMidHits >= kMidBox) would keep their own hits per lane.
    uint32_t incl = mine_n;
    incl += dpp_u<kDppRowShr1>(0u, incl);
    incl += dpp_u<kDppRowShr2>(0u, incl);
    incl += dpp_u<kDppRowShr4>(0u, incl);
    incl += dpp_u<kDppRowShr8>(0u, incl);
    const bool compact = ntot <= (uint32_t)kMidHits;
    if (compact) {
      uint32_t at = incl - mine_n;
      for (unsigned long long m = hits; m; m &= m - 1ull) my_hits[at++] = (uint16_t)(l16 + 16 * (__ffsll((long long)m) - 1));
    }
    wave_sync();
    // pass 2: the class vectors of the hits, weighted as the reference weights them
    float tri[CT];
#pragma unroll
    for (int c = 0; c < CT; c++) tri[c] = 0.0f;
    const float inv_bh = 1.0f / (float)bh;
    uint32_t round = 0u;
    while (true) {
      bool have;
      int i;
      if (compact) {
        const uint32_t e = round * 16u + (uint32_t)l16;
        have = e < ntot;
        i = have ? (int)my_hits[e] : 0;
      } else {
        have = hits != 0ull;
        i = have ? l16 + 16 * (__ffsll((long long)hits) - 1) : 0;   // pixel number inside the box, < 1024 (no hit: the box origin)
        hits &= hits - 1ull;
      }
      round++;
      if (__ballot(have) == 0ull) break;
      int ix = (int)(((float)i + 0.5f) * inv_bh);                          // i / bh, corrected below
      int iy = i - ix * bh;
      if (iy < 0) { ix--; iy += bh; } else if (iy >= bh) { ix++; iy -= bh; }
      const uint32_t hx = (uint32_t)(x0 + ix), hy = (uint32_t)(y0 + iy);
      float p[CT];
      load_row<CT, false>(probs + ((uint64_t)hx * ps0 + (uint64_t)hy * ps1), C, p);
      const float wt = (weights && have) ? weights[(uint64_t)hx * vH + hy] : 1.0f;
      float sum = 0.0f;
#pragma unroll
      for (int c = 0; c < CT; c++) if (c < C) sum = sum + p[c];
      if (have && sum > 0.5f) {                                  // Mesh.h:98
        const float wgt = w0 * wt;                               // :103
        if (KIND == SMESH_AGG_SUMMAX) {
          float best = p[0];
          int am = 0;
#pragma unroll
          for (int c = 1; c < CT; c++) if (c < C) if (p[c] > best) { best = p[c]; am = c; }
#pragma unroll
          for (int c = 0; c < CT; c++) if (c < C) tri[c] = (c == am) ? tri[c] + p[c] * wgt : tri[c];
        } else {
#pragma unroll
          for (int c = 0; c < CT; c++) if (c < C) tri[c] = tri[c] + p[c] * wgt;
        }
      }
    }
    // ---- the row-wide totals; lane l adds classes l, l + 16, l + 32 to the accumulator row
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int c = 0; c < CT; c++) {
      if (c < C) {
        const float t = row16_sum(tri[c]);
        if ((c & 15) == l16) { if (c < 16) m0 = t; else if (c < 32) m1 = t; else m2 = t; }
      }
    }
    if (on && ntot && !(SMESH_ABL(a.dbg) & 2)) {
      float* __restrict__ row = a.acc + (uint64_t)pid * C;
      if (l16 < C && m0 != 0.0f) unsafeAtomicAdd(&row[l16], m0);
      if (l16 + 16 < C && m1 != 0.0f) unsafeAtomicAdd(&row[l16 + 16], m1);
      if (CT > 32 && l16 + 32 < C && m2 != 0.0f) unsafeAtomicAdd(&row[l16 + 32], m2);
    }
  }
}


