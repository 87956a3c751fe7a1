// fusion.hip -- MeshAggregator on MI355X (gfx950): per-view histogram, segmented scatter-add, finalize.
//
// Replaces the reference's host-side fusion (citations relative to /root/reference):
//   include/semantic_meshes/fusion/Mesh.h:90-93   serial std::map histogram      -> k_hist
//   include/semantic_meshes/fusion/Mesh.h:94-106  OpenMP loop + per-primitive mutex -> k_scatter_tile
//   python/semantic_meshes/src/Fusion.cu:46-92    Summax / Sum / Mul aggregators  -> KIND template
//   python/semantic_meshes/include/Fusion.h:26-40 TensorConstructor copy+cast      -> k_gather_* (only
//                                                  when the caller's layout is not already contiguous)
//   Fusion.h:79-104 + Fusion.cu:47-49,67-69,79-82 get() functor chain             -> k_finalize_tile
//
// Data layout in HBM: accumulator float32[P][C] dense row-major (what get() returns and what the
// cross-GPU all-reduce sums); per-view histogram uint32[P] kept zero between add() calls.
//
// The scatter-add is HBM-bound (no MFMA): per view it must read 4*N (indices) + 4*N*C (probs) bytes
// and read-modify-write 2*4*C*T accumulator bytes (T = distinct primitives touched).  Design:
//   * one workgroup = one tile of TP consecutive pixels (images are y-fastest, so a tile is a run of
//     pixels down a column); the tile's probs (TP*C floats, contiguous in memory) are streamed with
//     16-byte coalesced loads into LDS,
//   * a wave ballot over "index differs from the previous pixel" splits the tile into same-primitive
//     runs (a triangle projects to vertically adjacent pixels); each run is reduced out of LDS by the
//     lanes that own its (run, class) elements,
//   * one global float atomic per (run, class), issued by consecutive lanes on consecutive addresses.
#include "common.hpp"

#include <cmath>
#include <new>

using namespace smesh;

namespace {

constexpr int kWave = 64;

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

__global__ void k_gather_probs(const float* __restrict__ in, int64_t s0, int64_t s1, int64_t s2,
                               float* __restrict__ out, uint64_t total, uint32_t H, uint32_t C) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const uint64_t pix = e / C;
  const uint32_t c = (uint32_t)(e - pix * C);
  const uint64_t x = pix / H, y = pix - x * H;
  out[e] = in[x * s0 + y * s1 + (int64_t)c * s2];
}

// ------------------------------------------------------------------------------------------------
// F1: per-view histogram count[v] = #pixels with index v (Mesh.h:90-93).  Same-index runs inside a
// wave are collapsed to one atomic by the run's first lane.
// ------------------------------------------------------------------------------------------------
__global__ void k_hist(const uint32_t* __restrict__ idx, uint32_t* __restrict__ count, uint64_t N, uint32_t P) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & (kWave - 1);
  const bool in = i < N;
  const uint32_t v = in ? idx[i] : 0xFFFFFFFFu;
  uint32_t prev = __shfl_up(v, 1);
  const bool head = in && (lane == 0 || v != prev);
  const unsigned long long heads = __ballot(head);
  const unsigned long long active = __ballot(in);
  if (head && v < P) {
    // run length = distance to the next head (or to the end of the active lanes)
    const unsigned long long later = lane == 63 ? 0ull : (heads >> (lane + 1));
    const int nactive = __popcll(active);  // active lanes are a prefix of the wave
    const int len = later ? (__ffsll((long long)later)) : (nactive - lane);
    atomicAdd(&count[v], (uint32_t)len);
  }
}

// Zero only the touched histogram entries (cheaper than a memset when P >> N).
__global__ void k_hist_clear(const uint32_t* __restrict__ idx, uint32_t* __restrict__ count, uint64_t N, uint32_t P) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const uint32_t v = idx[i];
  if (v < P) count[v] = 0u;
}

// ------------------------------------------------------------------------------------------------
// aggregator input maps (Fusion.cu:51-56 Summax, :70-73 Sum, :83-87 Mul)
// ------------------------------------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ float contribution(float p, float w) {
  if (KIND == SMESH_AGG_MUL) return logf(powf(p, w));  // LogProb of p^w
  return p * w;
}

// Opaque to the optimiser: the value must sit in VGPRs here, so the load that produced it cannot be sunk
// into a later conditional block (which would serialise the tile's loads one s_waitcnt at a time).
__device__ __forceinline__ void pin(float4& v) {
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
}

struct ScatterArgs {
  const uint32_t* idx;
  const float* probs;
  const float* weights;   // may be null
  const uint32_t* count;  // null when images_equal_weight == 0 (weight does not depend on the histogram)
  float* acc;
  uint64_t N;
  uint32_t P;
  uint32_t C;
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
// F2: segmented scatter-add, one tile of TP pixels per workgroup.
// ------------------------------------------------------------------------------------------------
template <int CT, int KIND, int TP>
__global__ __launch_bounds__(TP) void k_scatter_tile(ScatterArgs a) {
  const int C = CT > 0 ? CT : (int)a.C;
  constexpr int NW = TP / kWave;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int pfloats = (TP * C + 3) & ~3;
  float* sp = reinterpret_cast<float*>(smem);               // [TP*C] probs tile, flat copy of global
  float* sw = sp + pfloats;                                 // [TP]   per-pixel weight (0 = skip)
  uint32_t* sprim = reinterpret_cast<uint32_t*>(sw + TP);   // [TP]   primitive of run r
  int* sstart = reinterpret_cast<int*>(sprim + TP);         // [TP+1] first pixel of run r
  int* swave = sstart + TP + 1;                             // [NW]   heads per wave
  uint16_t* samax = reinterpret_cast<uint16_t*>(swave + NW + 1);  // [TP] arg-max class (Summax only)

  const int t = threadIdx.x;
  const int lane = t & (kWave - 1), wave = t / kWave;
  const uint64_t b0 = (uint64_t)blockIdx.x * TP;
  const int npx = (int)((a.N - b0) < (uint64_t)TP ? (a.N - b0) : (uint64_t)TP);
  const int nfl = npx * C;
  const float* __restrict__ src = a.probs + b0 * (uint64_t)C;

  // ---- stage 1: stream the tile's probs into LDS (16 B per lane, coalesced) -----------------
  if (npx == TP) {
    const float4* __restrict__ src4 = reinterpret_cast<const float4*>(src);
    float4* sp4 = reinterpret_cast<float4*>(sp);
    const int nvec = (TP * C) >> 2;  // TP is a multiple of 4
    if constexpr (CT > 0) {
      constexpr int KV = (CT + 3) / 4;  // vectors per thread
      float4 r[KV];
#pragma unroll
      for (int k = 0; k < KV; k++) {
        const int e = t + k * TP;
        r[k] = src4[e < nvec ? e : nvec - 1];  // clamped, unconditional: all KV loads are in flight together
      }
#pragma unroll
      for (int k = 0; k < KV; k++) pin(r[k]);  // keep the loads above the (conditional) LDS stores
#pragma unroll
      for (int k = 0; k < KV; k++) {
        const int e = t + k * TP;
        if (e < nvec) sp4[e] = r[k];
      }
    } else {
      for (int base = 0; base < nvec; base += 4 * TP) {
        float4 r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = base + t + k * TP;
          r[k] = src4[e < nvec ? e : nvec - 1];
        }
#pragma unroll
        for (int k = 0; k < 4; k++) pin(r[k]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int e = base + t + k * TP;
          if (e < nvec) sp4[e] = r[k];
        }
      }
    }
  } else {
    for (int e = t; e < nfl; e += TP) sp[e] = src[e];
  }

  // ---- per-pixel scalars ---------------------------------------------------------------------
  const bool in = t < npx;
  const uint32_t v = in ? a.idx[b0 + t] : 0xFFFFFFFFu;
  const float wt = (in && a.weights) ? a.weights[b0 + t] : 1.0f;
  const bool prim_ok = in && v < a.P;                 // Mesh.h:95
  float w = prim_ok ? pixel_weight(a, v, wt) : 0.0f;

  // run heads: a pixel starts a run when its primitive differs from the previous pixel's
  uint32_t prev = __shfl_up(v, 1);
  if (lane == 0 && t > 0 && in) prev = a.idx[b0 + t - 1];
  const bool head = in && (t == 0 || v != prev);
  const unsigned long long heads = __ballot(head);
  if (lane == 0) swave[wave] = __popcll(heads);
  __syncthreads();

  int run_base = 0, nruns = 0;
#pragma unroll
  for (int k = 0; k < NW; k++) {
    const int c = swave[k];
    if (k < wave) run_base += c;
    nruns += c;
  }
  if (head) {
    const int r = run_base + __popcll(heads & ((1ull << lane) - 1ull));
    sstart[r] = t;
    sprim[r] = v;
  }
  if (t == 0) sstart[nruns] = npx;

  // ---- stage 2: don't-care test on the float32 sequential class sum (Mesh.h:98) ---------------
  if (in) {
    const float* row = sp + t * C;
    float s = 0.0f;
    float best = row[0];
    int m = 0;
    for (int c = 0; c < C; c++) {
      const float p = row[c];
      s = s + p;
      if (KIND == SMESH_AGG_SUMMAX && p > best) { best = p; m = c; }  // first max (Fusion.cu:53)
    }
    if (!(s > 0.5f)) w = 0.0f;
    if (KIND == SMESH_AGG_SUMMAX) samax[t] = (uint16_t)m;
  }
  sw[t] = w;
  __syncthreads();

  // ---- stage 3: each lane owns (run, class) elements; reduce the run out of LDS, one atomic ----
  const int total = nruns * C;
  for (int e = t; e < total; e += TP) {
    const int r = e / C;
    const int c = e - r * C;
    const uint32_t prim = sprim[r];
    if (prim >= a.P) continue;
    const int j0 = sstart[r], j1 = sstart[r + 1];
    float sum = 0.0f;
    bool touched = false;
    for (int j = j0; j < j1; j++) {
      const float wj = sw[j];
      if (wj != 0.0f) {
        if (KIND == SMESH_AGG_SUMMAX) {
          if ((int)samax[j] == c) { sum += sp[j * C + c] * wj; touched = true; }
        } else {
          sum += contribution<KIND>(sp[j * C + c], wj);
          touched = true;
        }
      }
    }
    if (touched) unsafeAtomicAdd(&a.acc[(uint64_t)prim * C + c], sum);
  }
}

// ------------------------------------------------------------------------------------------------
// Fallback for class counts whose tile does not fit LDS: per-pixel weights, then a flat scatter.
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
  unsafeAtomicAdd(&a.acc[(uint64_t)a.idx[i] * a.C + c], contribution<KIND>(a.probs[e], w));
}

// ------------------------------------------------------------------------------------------------
// get(): load -> [Mul: / max element] -> L1 normalise -> NaN/Inf -> 0   (Fusion.h:79-104)
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

template <int KIND, int TP>
__global__ __launch_bounds__(TP) void k_finalize_tile(const float* __restrict__ acc, float* __restrict__ out,
                                                      uint64_t P, int C) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sp = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x;
  const uint64_t r0 = (uint64_t)blockIdx.x * TP;
  const int nrows = (int)((P - r0) < (uint64_t)TP ? (P - r0) : (uint64_t)TP);
  const int nfl = nrows * C;
  const float* __restrict__ src = acc + r0 * (uint64_t)C;
  float* __restrict__ dst = out + r0 * (uint64_t)C;
  const bool vec = (nrows == TP);
  if (vec) {
    const float4* src4 = reinterpret_cast<const float4*>(src);
    float4* sp4 = reinterpret_cast<float4*>(sp);
    for (int e = t; e < (nfl >> 2); e += TP) sp4[e] = src4[e];
  } else {
    for (int e = t; e < nfl; e += TP) sp[e] = src[e];
  }
  __syncthreads();
  if (t < nrows) finalize_row<KIND>(sp + t * C, C);
  __syncthreads();
  if (vec) {
    const float4* sp4 = reinterpret_cast<const float4*>(sp);
    float4* dst4 = reinterpret_cast<float4*>(dst);
    for (int e = t; e < (nfl >> 2); e += TP) dst4[e] = sp4[e];
  } else {
    for (int e = t; e < nfl; e += TP) dst[e] = sp[e];
  }
}

template <int KIND>
__global__ void k_finalize_rows(const float* __restrict__ acc, float* __restrict__ out, uint64_t P, int C) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const float* src = acc + p * C;
  float* dst = out + p * C;
  for (int c = 0; c < C; c++) dst[c] = src[c];
  finalize_row<KIND>(dst, C);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline int tile_pixels(uint32_t C) {
  if (C <= 48) return 256;
  if (C <= 100) return 128;
  if (C <= 220) return 64;
  return 0;  // fallback path
}

inline size_t tile_lds_bytes(int TP, uint32_t C) {
  const size_t pfloats = ((size_t)TP * C + 3) & ~(size_t)3;
  return pfloats * 4 + (size_t)TP * 4 /*sw*/ + (size_t)TP * 4 /*sprim*/ + (size_t)(TP + 1) * 4 /*sstart*/ +
         (size_t)(TP / kWave + 1) * 4 /*swave*/ + (size_t)TP * 2 /*samax*/ + 16;
}

template <int CT, int KIND>
int launch_tile_tp(const ScatterArgs& a, int TP, hipStream_t st) {
  const uint32_t grid = (uint32_t)div_up(a.N, TP);
  const size_t lds = tile_lds_bytes(TP, a.C);
  switch (TP) {
    case 256: hipLaunchKernelGGL((k_scatter_tile<CT, KIND, 256>), dim3(grid), dim3(256), lds, st, a); break;
    case 128: hipLaunchKernelGGL((k_scatter_tile<CT, KIND, 128>), dim3(grid), dim3(128), lds, st, a); break;
    default:  hipLaunchKernelGGL((k_scatter_tile<CT, KIND, 64>), dim3(grid), dim3(64), lds, st, a); break;
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

template <int KIND>
int launch_tile(const ScatterArgs& a, int TP, hipStream_t st) {
  // class counts of the benchmark configs get compile-time loops; everything else runs the same
  // kernel with a run-time C (the reference needs a rebuild with -DCLASSES_NUMS for each count)
  switch (a.C) {
    case 5:   return launch_tile_tp<5, KIND>(a, TP, st);
    case 19:  return launch_tile_tp<19, KIND>(a, TP, st);
    case 40:  return launch_tile_tp<40, KIND>(a, TP, st);
    default:  return launch_tile_tp<0, KIND>(a, TP, st);
  }
}

}  // namespace

struct smesh_aggregator {
  DeviceCtx* ctx = nullptr;
  uint64_t P = 0;
  uint32_t C = 0;
  int kind = 0;
  float iew = 0.5f;
  float* acc = nullptr;       // float32[P*C]
  uint32_t* count = nullptr;  // uint32[P], all zero between add() calls
  Scratch st_idx, st_probs, st_w;        // host->device staging
  Scratch nm_idx, nm_probs, nm_w;        // normalised (contiguous) copies
  Scratch fb_w, fb_amax;                 // fallback path scratch
  Scratch out_tmp;                       // get(): normalised result before the D2H copy
  std::mutex mu;
};

namespace {

int stage_in(DeviceCtx* ctx, Scratch& st, const void* host, size_t bytes, const void** dev) {
  SMESH_TRY(st.reserve(bytes));
  SMESH_HIP(hipMemcpyAsync(st.ptr, host, bytes, hipMemcpyHostToDevice, ctx->stream));
  *dev = st.ptr;
  return SMESH_OK;
}

size_t idx_itemsize(int dt) { return (dt == SMESH_IDX_U64 || dt == SMESH_IDX_I64) ? 8 : 4; }

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
    const uint64_t total = N * C;
    hipLaunchKernelGGL(k_gather_probs, dim3((uint32_t)div_up(total, 256)), dim3(256), 0, st, d_probs, ps[0], ps[1], ps[2],
                       static_cast<float*>(a->nm_probs.ptr), total, (uint32_t)H, C);
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

  // ---- F1 histogram (skipped when the weight does not depend on it) -------------------------
  const bool need_hist = a->iew != 0.0f;
  if (need_hist) {
    ProfScope prof(ctx, SMESH_PROF_FUSE_HIST);
    hipLaunchKernelGGL(k_hist, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, idx, a->count, N, (uint32_t)a->P);
    SMESH_HIP(hipGetLastError());
  }

  // ---- F2 scatter-add -------------------------------------------------------------------------
  ScatterArgs args;
  args.idx = idx; args.probs = probs; args.weights = weights;
  args.count = need_hist ? a->count : nullptr;
  args.acc = a->acc; args.N = N; args.P = (uint32_t)a->P; args.C = C; args.iew = a->iew;
  const int TP = tile_pixels(C);
  if (TP) {
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
    switch (a->kind) {
      case SMESH_AGG_SUM:    SMESH_TRY(launch_tile<SMESH_AGG_SUM>(args, TP, st)); break;
      case SMESH_AGG_SUMMAX: SMESH_TRY(launch_tile<SMESH_AGG_SUMMAX>(args, TP, st)); break;
      default:               SMESH_TRY(launch_tile<SMESH_AGG_MUL>(args, TP, st)); break;
    }
  } else {
    SMESH_TRY(a->fb_w.reserve(N * 4));
    SMESH_TRY(a->fb_amax.reserve(N * 4));
    float* wpix = static_cast<float*>(a->fb_w.ptr);
    uint32_t* amax = static_cast<uint32_t*>(a->fb_amax.ptr);
    ProfScope prof(ctx, SMESH_PROF_FUSE_SCATTER);
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
        hipLaunchKernelGGL(k_pixel_weight<SMESH_AGG_MUL>, g1, b, 0, st, args, wpix, amax);
        hipLaunchKernelGGL(k_scatter_flat<SMESH_AGG_MUL>, g2, b, 0, st, args, wpix, amax);
        break;
    }
    SMESH_HIP(hipGetLastError());
  }

  // ---- restore the all-zero histogram ----------------------------------------------------------
  if (need_hist) {
    if (a->P <= 2 * N) {
      SMESH_HIP(hipMemsetAsync(a->count, 0, a->P * 4, st));
    } else {
      hipLaunchKernelGGL(k_hist_clear, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, st, idx, a->count, N, (uint32_t)a->P);
      SMESH_HIP(hipGetLastError());
    }
  }
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

DeviceCtx* smesh_aggregator_ctx(smesh_aggregator* a) { return a->ctx; }
uint32_t smesh_aggregator_classes(smesh_aggregator* a) { return a->C; }
std::mutex& smesh_aggregator_mutex(smesh_aggregator* a) { return a->mu; }
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
  a->ctx = ctx; a->P = P; a->C = C; a->kind = kind; a->iew = iew;
  const size_t acc_bytes = (size_t)P * C * 4;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&a->acc), acc_bytes ? acc_bytes : 16);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&a->count), P ? P * 4 : 16);
  if (e == hipSuccess) e = hipMemsetAsync(a->acc, 0, acc_bytes, ctx->stream);   // Sum/Summax: 0; Mul: log 1 = 0
  if (e == hipSuccess) e = hipMemsetAsync(a->count, 0, P * 4, ctx->stream);
  if (e != hipSuccess) {
    if (a->acc) (void)hipFree(a->acc);
    if (a->count) (void)hipFree(a->count);
    delete a;
    return fail_hip(e, "aggregator allocation", __FILE__, __LINE__);
  }
  *out = a;
  return SMESH_OK;
}

int smesh_aggregator_destroy(smesh_aggregator_t* a) {
  if (!a) return SMESH_OK;
  (void)hipSetDevice(a->ctx->device);
  (void)hipStreamSynchronize(a->ctx->stream);
  (void)hipFree(a->acc);
  (void)hipFree(a->count);
  for (Scratch* s : {&a->st_idx, &a->st_probs, &a->st_w, &a->nm_idx, &a->nm_probs, &a->nm_w, &a->fb_w, &a->fb_amax, &a->out_tmp})
    s->release();
  delete a;
  return SMESH_OK;
}

int smesh_aggregator_reset(smesh_aggregator_t* a) {
  if (!a) return fail(SMESH_ERR_INVALID, "NULL aggregator");
  std::lock_guard<std::mutex> g(a->mu);
  std::lock_guard<std::recursive_mutex> lock(a->ctx->mu);
  SMESH_HIP(hipSetDevice(a->ctx->device));
  SMESH_HIP(hipMemsetAsync(a->acc, 0, (size_t)a->P * a->C * 4, a->ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_add(smesh_aggregator_t* a, const void* indices, int idx_dtype, const int64_t is[2], int imem,
                         const float* probs, const int64_t ps[3], int pmem,
                         const float* weights, const int64_t ws[2], int wmem, uint64_t W, uint64_t H) {
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
  SMESH_TRY(add_device(a, d_idx, idx_dtype, is, d_probs, ps, d_w, ws, W, H));
  // host buffers may be reused by the caller as soon as we return; pageable H2D copies have already
  // been consumed into the staging buffers, device inputs must stay valid until the stream drains
  if (imem == SMESH_MEM_DEVICE || pmem == SMESH_MEM_DEVICE || (weights && wmem == SMESH_MEM_DEVICE)) {
    // inputs are never retained after return (Fusion.h:45-47): wait for the kernels that read them
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
  }
  return SMESH_OK;
}

static int finalize_into(smesh_aggregator* a, float* d_out) {
  DeviceCtx* ctx = a->ctx;
  if (a->P == 0) return SMESH_OK;
  ProfScope prof(ctx, SMESH_PROF_FINALIZE);
  const int TP = tile_pixels(a->C);
  const int C = (int)a->C;
  if (TP) {
    const size_t lds = (((size_t)TP * C + 3) & ~(size_t)3) * 4;
    const dim3 g((uint32_t)div_up(a->P, TP));
#define SMESH_FIN(K)                                                                                          \
    switch (TP) {                                                                                             \
      case 256: hipLaunchKernelGGL((k_finalize_tile<K, 256>), g, dim3(256), lds, ctx->stream, a->acc, d_out, a->P, C); break; \
      case 128: hipLaunchKernelGGL((k_finalize_tile<K, 128>), g, dim3(128), lds, ctx->stream, a->acc, d_out, a->P, C); break; \
      default:  hipLaunchKernelGGL((k_finalize_tile<K, 64>), g, dim3(64), lds, ctx->stream, a->acc, d_out, a->P, C); break;  \
    }
    switch (a->kind) {
      case SMESH_AGG_SUM: SMESH_FIN(SMESH_AGG_SUM); break;
      case SMESH_AGG_SUMMAX: SMESH_FIN(SMESH_AGG_SUMMAX); break;
      default: SMESH_FIN(SMESH_AGG_MUL); break;
    }
#undef SMESH_FIN
  } else {
    const dim3 g((uint32_t)div_up(a->P, 256)), b(256);
    switch (a->kind) {
      case SMESH_AGG_SUM: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_SUM>, g, b, 0, ctx->stream, a->acc, d_out, a->P, C); break;
      case SMESH_AGG_SUMMAX: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_SUMMAX>, g, b, 0, ctx->stream, a->acc, d_out, a->P, C); break;
      default: hipLaunchKernelGGL(k_finalize_rows<SMESH_AGG_MUL>, g, b, 0, ctx->stream, a->acc, d_out, a->P, C); break;
    }
  }
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}

int smesh_aggregator_get(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (bytes == 0) return SMESH_OK;
  if (memkind == SMESH_MEM_DEVICE) {
    SMESH_TRY(finalize_into(a, out));
    SMESH_HIP(hipStreamSynchronize(ctx->stream));
    return SMESH_OK;
  }
  SMESH_TRY(a->out_tmp.reserve(bytes));
  SMESH_TRY(finalize_into(a, static_cast<float*>(a->out_tmp.ptr)));
  SMESH_HIP(hipMemcpyAsync(out, a->out_tmp.ptr, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_get_raw(smesh_aggregator_t* a, float* out, int memkind) {
  if (!a || !out) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (!bytes) return SMESH_OK;
  SMESH_HIP(hipMemcpyAsync(out, a->acc, bytes, memkind == SMESH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_set_raw(smesh_aggregator_t* a, const float* in, int memkind) {
  if (!a || !in) return fail(SMESH_ERR_INVALID, "NULL argument");
  std::lock_guard<std::mutex> g(a->mu);
  DeviceCtx* ctx = a->ctx;
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(ctx->device));
  const size_t bytes = (size_t)a->P * a->C * 4;
  if (!bytes) return SMESH_OK;
  SMESH_HIP(hipMemcpyAsync(a->acc, in, bytes, memkind == SMESH_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, ctx->stream));
  SMESH_HIP(hipStreamSynchronize(ctx->stream));
  return SMESH_OK;
}

int smesh_aggregator_raw_pointer(smesh_aggregator_t* a, void** ptr, uint64_t* n) {
  if (!a || !ptr) return fail(SMESH_ERR_INVALID, "NULL argument");
  // callers (the RCCL all-reduce) use this from another stream: make sure our work is done first
  SMESH_HIP(hipSetDevice(a->ctx->device));
  SMESH_HIP(hipStreamSynchronize(a->ctx->stream));
  *ptr = a->acc;
  if (n) *n = a->P * a->C;
  return SMESH_OK;
}

}  // extern "C"
