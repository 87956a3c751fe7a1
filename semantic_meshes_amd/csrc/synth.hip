// synth.hip -- synthetic class-probability images generated directly in HBM (benchmark utility).
//
// benchmark inputs must be resident in device memory before the timed region starts and
// must be reproducible on the CPU for parity checks.  The arithmetic (hash -> uniform -> u^4 -> normalise)
// avoids transcendental functions so that this kernel and oracle/smesh_oracle.cpp:smesh_synth_probs
// produce identical bits (this file is compiled with -ffp-contract=off).
#include "common.hpp"

#pragma clang fp contract(off)

using namespace smesh;

namespace {

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// one wave-lane per pixel keeps the class sum sequential (bit-exact with the CPU generator)
__global__ void k_synth_probs(float* __restrict__ out, uint64_t N, uint32_t C, uint64_t seed, uint32_t zthr) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  float* row = out + i * C;
  const uint64_t hz = splitmix64(seed ^ (0xD1B54A32D192ED03ull * (i + 1)));
  if ((uint32_t)(hz >> 40) < zthr) {
    for (uint32_t c = 0; c < C; c++) row[c] = 0.0f;
    return;
  }
  float s = 0.0f;
  for (uint32_t c = 0; c < C; c++) {
    const uint64_t h = splitmix64(seed + 0x632BE59BD9B4E019ull * (i * C + c + 1));
    const float u = (float)(uint32_t)(h >> 40) * (1.0f / 16777216.0f);
    const float u2 = u * u;
    const float q = u2 * u2 + 1e-4f;
    row[c] = q;
    s = s + q;
  }
  for (uint32_t c = 0; c < C; c++) row[c] = row[c] / s;
}

}  // namespace

extern "C" int smesh_synth_probs(float* out, uint64_t N, uint32_t C, uint64_t seed, float zero_fraction, int device,
                                 int memkind) {
  if (!out) return fail(SMESH_ERR_INVALID, "out is NULL");
  if (memkind != SMESH_MEM_DEVICE) return fail(SMESH_ERR_INVALID, "smesh_synth_probs writes device memory only");
  if (N == 0 || C == 0) return SMESH_OK;
  DeviceCtx* ctx;
  SMESH_TRY(get_ctx(device, &ctx));
  std::lock_guard<std::recursive_mutex> lock(ctx->mu);
  SMESH_HIP(hipSetDevice(device));
  const uint32_t zthr = (uint32_t)(zero_fraction * 16777216.0f);
  hipLaunchKernelGGL(k_synth_probs, dim3((uint32_t)div_up(N, 256)), dim3(256), 0, ctx->stream, out, N, C, seed, zthr);
  SMESH_HIP(hipGetLastError());
  return SMESH_OK;
}
