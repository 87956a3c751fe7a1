/* C99 client of include/smesh.h: what a non-Python binding (cgo, JNI, a C++ host) would do.  Compiled with gcc by
 * tests/test_abi.py (syntax check, and run against the CPU oracle without the device-only part) and built + run against
 * libsmesh_hip.so by tests/test_gpu_parity.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "smesh.h"

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int st_ = (call);                                                                        \
    if (st_ != SMESH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, st_, smesh_last_error()); return 1; } \
  } while (0)

int main(void) {
  /* two triangles forming a square in the plane z = 0, seen from z = +3 looking down -z (rotation = diag(1,-1,-1)) */
  const float verts[4 * 3] = {-1, -1, 0, 1, -1, 0, 1, 1, 0, -1, 1, 0};
  const int32_t faces[2 * 3] = {0, 1, 2, 0, 2, 3};
  enum { W = 64, H = 48, C = 3 };
  smesh_camera_t cam;
  memset(&cam, 0, sizeof cam);
  cam.rotation[0] = 1; cam.rotation[4] = -1; cam.rotation[8] = -1;
  cam.translation[2] = 3;
  cam.focal[0] = cam.focal[1] = 40; cam.principal[0] = W / 2.0; cam.principal[1] = H / 2.0;
  cam.width = W; cam.height = H;

  int ndev = 0;
  CHECK(smesh_device_count(&ndev));
#ifndef ABI_SMOKE_SKIP_DEVICE_COUNT   /* (the CPU oracle exports the same ABI and reports no device) */
  if (ndev < 1) { fprintf(stderr, "no device\n"); return 2; }
#endif
  smesh_renderer_t* r = NULL;
  CHECK(smesh_renderer_create_triangles(verts, 4, faces, 2, 0, &r));
  uint64_t P = 0;
  CHECK(smesh_renderer_num_primitives(r, &P));
  if (P != 2) return 3;
  uint32_t* idx = (uint32_t*)malloc(sizeof(uint32_t) * W * H);
  float* depth = (float*)malloc(sizeof(float) * W * H);
  float* probs = (float*)malloc(sizeof(float) * W * H * C);
  CHECK(smesh_renderer_render(r, &cam, idx, depth));
  long covered[2] = {0, 0}, background = 0;
  for (int i = 0; i < W * H; i++) {
    if (idx[i] == 0xFFFFFFFFu) { background++; if (!isinf(depth[i])) return 4; }
    else if (idx[i] < 2) { covered[idx[i]]++; if (fabsf(depth[i] - 3.0f) > 1e-4f) return 5; }
    else return 6;
  }
  if (covered[0] < 100 || covered[1] < 100 || background < 100) return 7;
  for (int i = 0; i < W * H; i++) { probs[i * C] = 0.7f; probs[i * C + 1] = 0.2f; probs[i * C + 2] = 0.1f; }

  smesh_aggregator_t* a = NULL;
  CHECK(smesh_aggregator_create(P, C, SMESH_AGG_SUM, 0.5f, 0, &a));
  const int64_t istr[2] = {H, 1}, pstr[3] = {H * C, C, 1};
  CHECK(smesh_aggregator_add(a, idx, SMESH_IDX_U32, istr, SMESH_MEM_HOST, probs, pstr, SMESH_MEM_HOST, NULL, NULL, SMESH_MEM_HOST, W, H));
  if (strlen(smesh_last_add_path()) == 0) return 12;           /* "scatter" / "image-records" (HIP library), "oracle" */
  CHECK(smesh_fuse_view(r, a, &cam, probs, NULL, SMESH_MEM_HOST));
  float out[2 * C];
  CHECK(smesh_aggregator_get(a, out, SMESH_MEM_HOST));
  for (int p = 0; p < 2; p++)
    if (fabsf(out[p * C] - 0.7f) > 1e-5f || fabsf(out[p * C + 1] - 0.2f) > 1e-5f || fabsf(out[p * C + 2] - 0.1f) > 1e-5f) return 8;
#ifndef ABI_SMOKE_SKIP_DEVICE_COUNT
  /* the two-call convention on the device: render_device() -> add_rendered(); a device buffer of the caller's, ordered against
   * the caller's stream (NULL = the legacy default stream) without host waits; then the same image as a COPY, found by content */
  {
    uint32_t* d_idx = NULL; float* d_depth = NULL; void* d_probs = NULL; int matched = -1;
    uint32_t* copy = (uint32_t*)malloc(sizeof(uint32_t) * W * H);
    CHECK(smesh_aggregator_reset(a));
    CHECK(smesh_device_malloc(0, sizeof(float) * W * H * C, &d_probs));
    CHECK(smesh_memcpy(d_probs, probs, sizeof(float) * W * H * C, SMESH_MEM_DEVICE, SMESH_MEM_HOST, 0));
    CHECK(smesh_renderer_render_device(r, &cam, &d_idx, &d_depth));
    CHECK(smesh_stream_wait(0, NULL));
    CHECK(smesh_aggregator_add_rendered(a, r, d_idx, (const float*)d_probs, pstr, SMESH_MEM_DEVICE, NULL, NULL, SMESH_MEM_HOST, W, H));
    CHECK(smesh_stream_release(0, NULL));
    CHECK(smesh_renderer_seal_render(r, d_idx));                         /* before the content leaves the library */
    CHECK(smesh_memcpy(copy, d_idx, sizeof(uint32_t) * W * H, SMESH_MEM_HOST, SMESH_MEM_DEVICE, 0));
    if (memcmp(copy, idx, sizeof(uint32_t) * W * H) != 0) return 11;
    CHECK(smesh_aggregator_add_matched(a, r, copy, SMESH_IDX_U32, istr, SMESH_MEM_HOST, probs, pstr, SMESH_MEM_HOST, NULL, NULL, SMESH_MEM_HOST, W, H, &matched));
    if (matched != 1) return 12;
    copy[0] ^= 1u;                                                       /* no longer a copy of the render */
    CHECK(smesh_aggregator_add_matched(a, r, copy, SMESH_IDX_U32, istr, SMESH_MEM_HOST, probs, pstr, SMESH_MEM_HOST, NULL, NULL, SMESH_MEM_HOST, W, H, &matched));
    if (matched != 0) return 13;
    CHECK(smesh_synchronize(0));
    CHECK(smesh_renderer_release_image(r, d_idx, d_depth));
    CHECK(smesh_device_free(0, d_probs));
    free(copy);
    CHECK(smesh_aggregator_get(a, out, SMESH_MEM_HOST));
    for (int p = 0; p < 2; p++)
      if (fabsf(out[p * C] - 0.7f) > 1e-5f || fabsf(out[p * C + 1] - 0.2f) > 1e-5f) return 14;
    /* timing hooks: one bracketed region, one launch of the fusion kernel, one view */
    double ms = 0; uint64_t regions = 0, launches = 0, views = 0;
    CHECK(smesh_profile_reset(0));
    CHECK(smesh_profile_enable(0, 1 << SMESH_PROF_FUSE_SCATTER));
    CHECK(smesh_fuse_view(r, a, &cam, probs, NULL, SMESH_MEM_HOST));
    CHECK(smesh_profile_enable(0, 0));
    CHECK(smesh_profile_read_ex(0, SMESH_PROF_FUSE_SCATTER, &ms, &regions, &launches, &views));
    if (regions != 1 || launches != 1 || views != 1 || !(ms > 0)) return 15;
    /* a communicator of one rank: the all-reduce leaves the accumulator as it is */
    uint8_t id[SMESH_COMM_ID_BYTES]; smesh_comm_t* comm = NULL; float before[2 * C], after[2 * C];
    CHECK(smesh_aggregator_get_raw(a, before, SMESH_MEM_HOST));
    CHECK(smesh_comm_unique_id(id));
    CHECK(smesh_comm_create(0, 1, 0, id, &comm));
    { smesh_comm_t* cs[1]; smesh_aggregator_t* as[1]; cs[0] = comm; as[0] = a; CHECK(smesh_allreduce(cs, as, 1)); }
    CHECK(smesh_aggregator_get_raw(a, after, SMESH_MEM_HOST));
    if (memcmp(before, after, sizeof before) != 0) return 16;
    CHECK(smesh_comm_destroy(comm));
  }
#endif
  /* invalid arguments come back as SMESH_ERR_INVALID with a message, never as a crash or an exit() */
  if (smesh_aggregator_add(a, idx, 99, istr, SMESH_MEM_HOST, probs, pstr, SMESH_MEM_HOST, NULL, NULL, SMESH_MEM_HOST, W, H) != SMESH_ERR_INVALID) return 9;
  if (strlen(smesh_last_error()) == 0) return 10;
  CHECK(smesh_aggregator_destroy(a));
  CHECK(smesh_renderer_destroy(r));
  free(idx); free(depth); free(probs);
  printf("abi smoke ok: %ld + %ld covered pixels, backend %s\n", covered[0], covered[1], smesh_backend());
  return 0;
}
