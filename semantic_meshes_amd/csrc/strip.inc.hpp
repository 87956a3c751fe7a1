// strip.inc.hpp -- runs, links and groups of one 4 x 16-pixel strip of an index image (one wave); shared by the histogram /
// scatter-add kernels (fusion.hip) and the image-records builder (image_records.hip).  Included inside an anonymous namespace,
// after fuse_tri.inc.hpp (kWave, wave_sync).
//
// A strip is 4 columns x 16 rows = one wave; lane l owns pixel (cx = l / 16, ty = l % 16) of the y-fastest image.
constexpr int kSX = 4;
constexpr int kTY = 16;
constexpr int kNone = 255;

// Wave-private LDS bookkeeping (no workgroup barrier is ever needed: the workgroup IS one wave).
struct StripLists {
  uint32_t sv[kWave];      // primitive of pixel l (0xFFFFFFFF outside the image)
  uint8_t shead[kWave];    // head lane of the run pixel l belongs to
  uint8_t lmatch[kWave];   // head lane of the first same-primitive run in the column to the left (or kNone)
  uint8_t rmatch[kWave];   // ... to the right
  uint8_t child[kWave];    // head lane -> next run of the chain (mutual match), or kNone
  uint8_t slen[kWave];     // head lane -> pixels in the run
  uint8_t groot[kWave];    // group g -> head lane of the chain's first (leftmost) run
  uint8_t amax[kWave];     // arg-max class of pixel l (Summax only)
  float sw[kWave];         // weight of pixel l (0 = contributes nothing)
};

struct StripRuns {
  bool head;       // this lane starts a run
  int hl;          // head lane of my run
  int len;         // pixels in my run (valid on head lanes)
  bool root;       // head lane of the first run of a chain with a valid primitive
  int gidx;        // root lanes: dense index of my group in [0, G)
  int G;           // groups in the strip
};

// Builds runs, links and groups from each lane's primitive id `v` (0xFFFFFFFF outside the image).
__device__ __forceinline__ StripRuns build_strip(StripLists& L, uint32_t v, uint32_t P, int l) {
  StripRuns r;
  const int ty = l & (kTY - 1), cx = l / kTY;
  // ---- runs: a pixel starts a run when it is the top of a column segment or differs from the pixel above
  L.sv[l] = v;
  const uint32_t prev = __shfl_up(v, 1);
  r.head = (ty == 0) || (v != prev);
  const unsigned long long heads = __ballot(r.head);
  const unsigned long long upto = (2ull << l) - 1ull;   // bits 0..l (l = 63: wraps to all ones)
  r.hl = 63 - __clzll((long long)(heads & upto));      // lane 0 is always a head
  const unsigned long long later = heads & ~upto;
  const int next = later ? (__ffsll((long long)later) - 1) : kWave;
  r.len = next - l;
  L.shead[l] = (uint8_t)r.hl;
  L.slen[l] = (uint8_t)r.len;
  wave_sync();
  // ---- first same-primitive run in the neighbouring columns (8-connectivity)
  const bool valid = v < P;
  int lp = kNone, rc = kNone;
  if (r.head && valid) {
    const int lo = max(ty - 1, 0), hi = min(ty + r.len, kTY - 1);
    if (cx < kSX - 1) {
      const int q0 = (cx + 1) * kTY;
      for (int y = lo; y <= hi; y++)
        if (L.sv[q0 + y] == v) { rc = L.shead[q0 + y]; break; }
    }
    if (cx > 0) {
      const int q0 = (cx - 1) * kTY;
      for (int y = lo; y <= hi; y++)
        if (L.sv[q0 + y] == v) { lp = L.shead[q0 + y]; break; }
    }
  }
  L.lmatch[l] = (uint8_t)lp;
  L.rmatch[l] = (uint8_t)rc;
  wave_sync();
  // ---- a link exists only when both runs chose each other, so chains never fork
  int child = kNone;
  r.root = false;
  if (r.head && valid) {
    if (rc != kNone && L.lmatch[rc] == l) child = rc;
    r.root = !(lp != kNone && L.rmatch[lp] == l);
  }
  L.child[l] = (uint8_t)child;
  const unsigned long long roots = __ballot(r.root);
  r.G = __popcll(roots);
  r.gidx = __popcll(roots & ((1ull << l) - 1ull));
  if (r.root) L.groot[r.gidx] = (uint8_t)l;
  wave_sync();
  return r;
}

