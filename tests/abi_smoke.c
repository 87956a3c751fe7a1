/* C99 client of include/smesh.h: what a non-Python binding (cgo, JNI, a C++ host) would do.  Compiled with gcc by
 * tests/test_abi.py (syntax only, no GPU) and built + run against libsmesh_hip.so by tests/test_gpu_parity.py. */
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
  CHECK(smesh_fuse_view(r, a, &cam, probs, NULL, SMESH_MEM_HOST));
  float out[2 * C];
  CHECK(smesh_aggregator_get(a, out, SMESH_MEM_HOST));
  for (int p = 0; p < 2; p++)
    if (fabsf(out[p * C] - 0.7f) > 1e-5f || fabsf(out[p * C + 1] - 0.2f) > 1e-5f || fabsf(out[p * C + 2] - 0.1f) > 1e-5f) return 8;
  /* invalid arguments come back as SMESH_ERR_INVALID with a message, never as a crash or an exit() */
  if (smesh_aggregator_add(a, idx, 99, istr, SMESH_MEM_HOST, probs, pstr, SMESH_MEM_HOST, NULL, NULL, SMESH_MEM_HOST, W, H) != SMESH_ERR_INVALID) return 9;
  if (strlen(smesh_last_error()) == 0) return 10;
  CHECK(smesh_aggregator_destroy(a));
  CHECK(smesh_renderer_destroy(r));
  free(idx); free(depth); free(probs);
  printf("abi smoke ok: %ld + %ld covered pixels, backend %s\n", covered[0], covered[1], smesh_backend());
  return 0;
}
