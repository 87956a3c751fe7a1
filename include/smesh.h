/*
 * smesh.h -- C ABI of the MI355X-native project-and-fuse hot path of semantic-meshes.
 *
 * This is the drop-in boundary: every entry point below is what a
 * binding for the reference's Python modules `semantic_meshes.render` / `semantic_meshes.fusion`
 * would call instead of the Boost.Python + template-tensors code.  Plain pointers and sizes
 * only; no torch / numpy / C++ types cross this boundary.
 *
 * Two shared libraries export this same ABI:
 *   semantic_meshes_amd/csrc/libsmesh_hip.so  -- the product: HIP kernels for gfx950 (MI355X)
 *   oracle/libsmesh_oracle.so                 -- TEST INFRASTRUCTURE ONLY: scalar CPU restatement
 *
 * Conventions (reference citations are relative to /root/reference):
 *   - images are (W,H[,C]) row-major with y fastest: element (x,y) lives at x*H + y
 *     (python/semantic_meshes/include/Renderer.h:29,32)
 *   - primitive index images are uint32, background 0xFFFFFFFF; depth float32, background +inf
 *     (include/semantic_meshes/render/TriangleRenderer.h:75-78; Renderer.h:19-23)
 *   - every function returns an int status; 0 = ok.  SMESH_ERR_INVALID maps to Python ValueError
 *     (std::invalid_argument in the reference: include/semantic_meshes/fusion/Mesh.h:68-74,
 *     python/semantic_meshes/src/Fusion.cu:122-125), everything else to RuntimeError.
 *   - strides are in ELEMENTS, not bytes, and must be >= 0.
 */
#ifndef SMESH_H
#define SMESH_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------------------ */
#define SMESH_OK            0
#define SMESH_ERR_INVALID   1   /* bad argument / shape / dtype  -> ValueError            */
#define SMESH_ERR_RUNTIME   2   /* HIP / RCCL / allocation failure -> RuntimeError         */
#define SMESH_ERR_NODEVICE  3   /* no usable GPU                 -> RuntimeError           */

/* ---- enums -------------------------------------------------------------------------------- */
/* index dtypes accepted by add(): python/semantic_meshes/include/Common.h:5-12 */
#define SMESH_IDX_U32 0
#define SMESH_IDX_I32 1
#define SMESH_IDX_U64 2
#define SMESH_IDX_I64 3

/* where a caller buffer lives: Common.h:10,19,28 (mem::HOST, mem::DEVICE) */
#define SMESH_MEM_HOST   0
#define SMESH_MEM_DEVICE 1

/* aggregator kinds: python/semantic_meshes/src/Fusion.cu:46-64 (Summax), :66-76 (Sum), :78-92 (Mul) */
#define SMESH_AGG_SUM    0
#define SMESH_AGG_SUMMAX 1
#define SMESH_AGG_MUL    2

/* ---- camera POD ---------------------------------------------------------------------------- */
/* Replaces semantic_meshes::Camera (include/semantic_meshes/render/Camera.h:7-15) as filled by the
 * Python ctor (python/semantic_meshes/include/Camera.h:16-57): float32 rigid world->camera
 * transform Xc = R*X + t, double per-axis focal lengths / principal point, resolution (W,H). */
typedef struct smesh_camera {
  float    rotation[9];      /* row-major 3x3 */
  float    translation[3];
  double   focal[2];         /* fx, fy (a scalar-f PinholeFC is passed as fx == fy) */
  double   principal[2];     /* cx, cy */
  uint64_t width;            /* resolution(0) */
  uint64_t height;           /* resolution(1) */
} smesh_camera_t;

typedef struct smesh_renderer   smesh_renderer_t;
typedef struct smesh_aggregator smesh_aggregator_t;

/* ---- library / device ---------------------------------------------------------------------- */
/* Returns a static string naming the backend: "hip-gfx950" or "oracle-cpu". */
const char* smesh_backend(void);
/* Thread-local message describing the last non-zero status returned on this thread. */
const char* smesh_last_error(void);
/* Number of visible GPUs (oracle: 0). */
int smesh_device_count(int* count);
/* Run-time options by name (the environment supplies the defaults).  "group_pipeline" (0 / 1; default 1, SMESH_GROUP_PIPELINE):
 * smesh_fuse_views rasterises group g + 1 of a batch on a second stream beside the fusion of group g -- same kernels, same inputs,
 * same results.  A harness that needs ONE kernel's own duration (a roofline) turns it off for that measurement. */
int smesh_set_option(const char* name, int64_t value);
int smesh_get_option(const char* name, int64_t* value);
/* Block until all work queued on `device` by this library has finished. */
int smesh_synchronize(int device);

/* Stream ordering (new: the reference moves every input with a synchronous cudaMemcpy on the null stream,
 * python/semantic_meshes/include/Fusion.h:35-37, which orders it after the producer implicitly; this library's streams are
 * non-blocking).  smesh_stream_wait orders the library's work on `device` after everything queued so far on `producer_stream`
 * (a hipStream_t; NULL = the legacy default stream) without blocking the host: call it before handing over DEVICE buffers that
 * another framework wrote on its own stream.  smesh_stream_handle returns the library's main hipStream_t so that a consumer can
 * order its own work after the library's (smesh_synchronize is the blocking alternative). */
int smesh_stream_wait(int device, void* producer_stream);
/* The other direction, for the asynchronous entry points (smesh_fuse_view(s), smesh_aggregator_add_rendered / _add_matched with DEVICE
 * images): `consumer_stream` (NULL = the legacy default stream) waits for everything the library has queued so far, so that the
 * owner of a DEVICE buffer can overwrite or free it on that stream without a host synchronisation. */
int smesh_stream_release(int device, void* consumer_stream);
int smesh_stream_handle(int device, void** stream);
/* Completion tokens (new): smesh_token_record marks the library's main stream of `device` behind everything queued so far -- no host
 * wait; smesh_token_done sets *done = 1 once the device has passed the mark (the token is then spent: do not query it again) and 0 while
 * it has not.  The asynchronous entry points (smesh_fuse_view(s), smesh_aggregator_add_async / _add_rendered / _add_matched) read their
 * DEVICE images after they return: a host layer that cannot order the owner's stream behind those reads (smesh_stream_release) keeps
 * the images alive until a token recorded after the call is done.  semantic_meshes_amd/fusion.py does both. */
int smesh_token_record(int device, uint64_t* token);
int smesh_token_done(int device, uint64_t token, int* done);
/* Time stamps on the library's main stream (new; a harness's view of where a job's device time went without a profiler):
 * smesh_stream_mark records mark `id` (0 .. 15) behind everything queued so far -- no host synchronisation;
 * smesh_stream_mark_elapsed waits for mark `to` and returns the device milliseconds between the two marks. */
#define SMESH_STREAM_MARKS 16
int smesh_stream_mark(int device, int id);
int smesh_stream_mark_elapsed(int device, int from, int to, double* ms);

/* ---- triangle renderer --------------------------------------------------------------------- */
/* Replaces TriangleRenderer::TriangleRenderer (include/semantic_meshes/render/TriangleRenderer.h:30-39):
 * uploads float32[V,3] vertices and int32[F,3] faces; primitive id == face ordinal (:57-60). */
int smesh_renderer_create_triangles(const float* vertices, uint64_t num_vertices,
                                    const int32_t* faces, uint64_t num_faces,
                                    int device, smesh_renderer_t** out);
/* Replaces TexturedTriangleRenderer::TexturedTriangleRenderer
 * (include/semantic_meshes/render/TexturedTriangleRenderer.h:87-182): texel primitives whose
 * per-triangle resolution comes from the max projected area over `cameras`. */
int smesh_renderer_create_texels(const float* vertices, uint64_t num_vertices,
                                 const int32_t* faces, uint64_t num_faces,
                                 const smesh_camera_t* cameras, uint64_t num_cameras,
                                 float texels_per_pixel, int device, smesh_renderer_t** out);
int smesh_renderer_destroy(smesh_renderer_t* r);
/* Replaces getPrimitivesNum() (TriangleRenderer.h:41-44, TexturedTriangleRenderer.h:184-187). */
int smesh_renderer_num_primitives(const smesh_renderer_t* r, uint64_t* out);
/* Texel renderers only: copies the (possibly re-ordered, TexturedTriangleRenderer.h:129-146) faces,
 * per-triangle texture resolution and first texel index to host arrays of F*3 / F / F entries.
 * Any pointer may be NULL. */
int smesh_renderer_texel_layout(const smesh_renderer_t* r, int32_t* faces_out,
                                uint32_t* resolution_out, uint32_t* first_texel_out);

/* Replaces Renderer<T>::render (python/semantic_meshes/include/Renderer.h:25-43) =
 * clear (TriangleRenderer.h:75-78) + rasterize (:81-88) + AoS->SoA split (Renderer.h:32-35).
 * Writes uint32[W,H] indices and float32[W,H] depth to HOST memory; depth_out may be NULL. */
int smesh_renderer_render(smesh_renderer_t* r, const smesh_camera_t* camera,
                          uint32_t* indices_out, float* depth_out);
/* Same, but the two planes stay in DEVICE memory owned by the renderer (the DLPack capsules of
 * Renderer.h:37-38).  The returned pointers stay valid until smesh_renderer_release_image() is
 * called on them or the renderer is destroyed; they can be passed straight to
 * smesh_aggregator_add(..., SMESH_MEM_DEVICE). */
int smesh_renderer_render_device(smesh_renderer_t* r, const smesh_camera_t* camera,
                                 uint32_t** indices_dev, float** depth_dev);
int smesh_renderer_release_image(smesh_renderer_t* r, void* indices_dev, void* depth_dev);

/* ---- mesh aggregator ----------------------------------------------------------------------- */
/* Replaces construct1/2/3 + ModelAggregator ctor (python/semantic_meshes/src/Fusion.cu:120-138;
 * include/semantic_meshes/fusion/Mesh.h:57-63).  num_classes is a run-time value here. */
int smesh_aggregator_create(uint64_t num_primitives, uint32_t num_classes, int kind,
                            float images_equal_weight, int device, smesh_aggregator_t** out);
int smesh_aggregator_destroy(smesh_aggregator_t* a);
/* Replaces ModelAggregator::reset (Mesh.h:119-122). */
int smesh_aggregator_reset(smesh_aggregator_t* a);

/* Replaces ModelAggregator::add1/add2 (python/semantic_meshes/include/Fusion.h:42-64) and
 * ModelAggregator::add (Mesh.h:65-107,109-117).
 *   indices: (W,H) image of `idx_dtype`, element strides idx_strides[2]
 *   probs:   (W,H,C) float32, element strides probs_strides[3]
 *   weights: (W,H) float32 or NULL (== all ones, Mesh.h:109-117), element strides weights_strides[2]
 * All three must have the same (W,H) -- the caller passes one (W,H); the Python layer raises
 * ValueError on mismatching shapes like Mesh.h:68-74.  Each buffer has its own memkind: the usual
 * call (python/scripts/colorize_cityscapes_mesh.py:65-67) passes render()'s DEVICE indices with
 * HOST network output. */
int smesh_aggregator_add(smesh_aggregator_t* a,
                         const void* indices, int idx_dtype, const int64_t idx_strides[2], int idx_memkind,
                         const float* probs, const int64_t probs_strides[3], int probs_memkind,
                         const float* weights, const int64_t weights_strides[2], int weights_memkind,
                         uint64_t width, uint64_t height);

/* smesh_aggregator_add that does not wait for the kernels reading DEVICE images (HOST images are still consumed before it
 * returns, Fusion.h:45-47): keep device images valid until smesh_stream_release / smesh_synchronize, like
 * smesh_aggregator_add_rendered.  New: lets consecutive add() calls on images the library did not render overlap -- the
 * per-primitive records of call k+1 (image_records.hip) are built on the main stream while call k is fused on a second one. */
int smesh_aggregator_add_async(smesh_aggregator_t* a,
                               const void* indices, int idx_dtype, const int64_t idx_strides[2], int idx_memkind,
                               const float* probs, const int64_t probs_strides[3], int probs_memkind,
                               const float* weights, const int64_t weights_strides[2], int weights_memkind,
                               uint64_t width, uint64_t height);
/* add() for a batch of `n` images of one size, in order (new functionality: the reference adds one image per call, Mesh.h:65-107).
 * Same sums as n smesh_aggregator_add_async calls -- per accumulator row the same float32 additions in the same order -- but dense
 * uint32 / int32 index planes and dense class vectors in DEVICE memory share their launches in groups of up to eight (one launch per
 * record pass, one triangle-order fusion launch per group).  One dtype, one set of strides and one memory kind for all images;
 * `weights` NULL or one pointer per image.  Asynchronous for device images (smesh_synchronize / smesh_token_*). */
int smesh_aggregator_add_many(smesh_aggregator_t* a, uint64_t n, const void* const* indices, int idx_dtype, const int64_t idx_strides[2],
                              int idx_memkind, const float* const* probs, const int64_t probs_strides[3], int probs_memkind,
                              const float* const* weights, const int64_t weights_strides[2], int weights_memkind, uint64_t W, uint64_t H);

/* Replaces ModelAggregator::get (Fusion.h:72-76; Mesh.h:131-132) with the output functor chain of
 * the chosen aggregator (Fusion.cu:47-49 / 67-69 / 79-82; Fusion.h:79-104): writes float32[P,C]. */
int smesh_aggregator_get(smesh_aggregator_t* a, float* out, int memkind);

/* smesh_aggregator_get for the rows [row_lo, row_hi) only (row_lo a multiple of 4): writes float32[row_hi - row_lo, C].
 * New: what a rank calls after smesh_reduce_scatter. */
int smesh_aggregator_get_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, float* out, int memkind);

/* Un-normalised accumulator state float32[P,C] (Sum/Summax: weighted sums; Mul: log-domain sums).
 * New functionality: this is what is summed across GPUs before get(). */
int smesh_aggregator_get_raw(smesh_aggregator_t* a, float* out, int memkind);
int smesh_aggregator_set_raw(smesh_aggregator_t* a, const float* in, int memkind);
/* Rows [row_lo, row_hi) of ONE PLANE of the raw state as dense float32[row_hi - row_lo, C]: plane 0 = the accumulator as stored (Mul:
 * the hi plane, not folded), plane 1 = Mul's lo plane (a Mul element's value is hi + lo).  What a host-side exchange of a row range
 * moves (semantic_meshes_amd/distributed.py: several ranks sharing one GPU exchange through gloo). */
int smesh_aggregator_get_raw_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, int plane, float* out, int memkind);
int smesh_aggregator_set_raw_rows(smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi, int plane, const float* in, int memkind);
/* In-place access for the RCCL all-reduce: the accumulator as it lives in device memory, P rows of
 * `row_stride` floats (row_stride >= C, padding is zero and stays zero under a sum), num_floats =
 * P * row_stride.  Oracle: host pointer, row_stride == C. */
int smesh_aggregator_raw_pointer(smesh_aggregator_t* a, void** ptr, uint64_t* num_floats);
int smesh_aggregator_row_stride(smesh_aggregator_t* a, uint32_t* row_stride);

/* ---- multi-GPU: one sum all-reduce of the raw accumulators over RCCL / xGMI (new) -------------- */
/* Views are independent and every aggregator is a sum in some domain (python/semantic_meshes/src/Fusion.cu:46-92), so each GPU
 * fuses its own views into a private accumulator and ONE ncclAllReduce(float32, sum) of the raw buffers precedes get().
 * A communicator spans `nranks` GPUs.  One process per GPU: rank 0 calls smesh_comm_unique_id, the host code hands the 128 bytes
 * to the other ranks (any side channel), every rank calls smesh_comm_create (collective, blocking).  One process driving several
 * GPUs: smesh_comm_create_all fills `out[0 .. ndev-1]`.  RCCL is loaded at run time; without it these return SMESH_ERR_RUNTIME. */
typedef struct smesh_comm smesh_comm_t;
#define SMESH_COMM_ID_BYTES 128
int smesh_comm_unique_id(uint8_t id[SMESH_COMM_ID_BYTES]);
int smesh_comm_create(int device, int nranks, int rank, const uint8_t id[SMESH_COMM_ID_BYTES], smesh_comm_t** out);
int smesh_comm_create_all(const int* devices, int ndev, smesh_comm_t** out);
int smesh_comm_destroy(smesh_comm_t* c);
int smesh_comm_rank(const smesh_comm_t* c, int* rank, int* nranks);
/* Sums the raw accumulators of all ranks in place in HBM.  `n` = the (communicator, aggregator) pairs THIS process holds: 1 in
 * the one-process-per-GPU layout, ndev after smesh_comm_create_all.  Asynchronous: enqueued on each device's library stream
 * right behind the fusion kernels already queued there; smesh_aggregator_get / smesh_synchronize order after it.  Every rank's
 * aggregator must have the same num_primitives, num_classes and kind. */
int smesh_allreduce(smesh_comm_t* const* comms, smesh_aggregator_t* const* aggs, int n);
/* Opt-in alternative to smesh_allreduce: ONE in-place ncclReduceScatter (half the bytes per xGMI link).  Afterwards rank r holds the
 * sum over all ranks in rows [*row_lo, *row_hi) of its accumulator: q = floor(P / nranks / 4) * 4 rows per rank (every slice starts
 * on a 16-byte boundary), the < 5 * nranks rows beyond nranks * q are all-reduced and counted into the last rank's range.  The other
 * rows keep the rank's own partial sums, so only smesh_aggregator_get_rows inside [*row_lo, *row_hi) is meaningful afterwards: each
 * rank normalises (Fusion.h:79-104) its own slice, and until smesh_aggregator_reset / _set_raw every other use of the accumulator
 * (get, get_raw, add, fuse_view, a second exchange) returns SMESH_ERR_INVALID.  Asynchronous like smesh_allreduce. */
int smesh_reduce_scatter(smesh_comm_t* c, smesh_aggregator_t* a, uint64_t* row_lo, uint64_t* row_hi);
/* ---- sharded jobs: exchange under the fusion (new; no reference counterpart -- the reference is single-GPU) ------- */
/* smesh_fuse_views cut by ACCUMULATOR ROW RANGE, so that the exchange of the rows that are already final runs beside the fusion of
 * the rest (the one collective of the sharded path would otherwise be fully exposed at the end of a rank's views).
 * smesh_fuse_views_begin rasterises all `n` views (n <= 32: their per-triangle records and index planes stay resident), then fuses
 * part 0 of `nparts`: for ALL n views, in order, only the triangles whose rows lie in [*row_lo, *row_hi) -- whole 64-row blocks,
 * part p = blocks [B p / nparts, B (p+1) / nparts) of the B = ceil(P / 64).  smesh_fuse_views_continue(part = 1, 2 ... nparts-1, in
 * that order) fuses the next range and returns it.  When a call returns (asynchronously, like smesh_fuse_views) the rows it names hold
 * everything the n views contribute to them: hand them to smesh_allreduce_rows and go on with the next part.  Per row the additions
 * are those of smesh_fuse_views on the same views, in its order (primitives wider than 8 x 8 pixels of at most 16 x 16: float atomics,
 * all with part 0).  Keep the DEVICE images valid until the last part has been queued.  Where rows are not in triangle order or the
 * views cannot be held (texel renderers, re-ordered meshes, HOST images, n > 32, class counts beyond the triangle-order kernels)
 * part 0 does the whole job -- rows [0, P) -- and the later parts are empty (*row_lo == *row_hi). */
int smesh_fuse_views_begin(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cameras, uint64_t n,
                           const float* const* probs, const float* const* weights, int memkind, int nparts,
                           uint64_t* row_lo, uint64_t* row_hi);
int smesh_fuse_views_continue(smesh_renderer_t* r, smesh_aggregator_t* a, int part, uint64_t* row_lo, uint64_t* row_hi);
/* smesh_allreduce for the rows [row_lo, row_hi) only, on the library's EXCHANGE stream: starts when what is queued on the main stream
 * so far has finished (an event, no host wait) and runs beside what the main stream is given next.  The main stream waits for it
 * (device-side) at the next entry point that touches the accumulator -- get(), add(), reset(), smesh_allreduce ... -- or at
 * smesh_exchange_join.  Every rank issues the same sequence of ranges.  Mul aggregators exchange their (hi, lo) float32 pairs as
 * float64 (twice the bytes; a float32 sum of log-domain partial sums misses 1e-5 on get()), here and in smesh_allreduce. */
int smesh_allreduce_rows(smesh_comm_t* c, smesh_aggregator_t* a, uint64_t row_lo, uint64_t row_hi);
int smesh_exchange_join(smesh_aggregator_t* a);
/* Blocking reduction of `n` <= 64 host doubles over the communicator (op 0 = sum, 2 = max): a harness's barrier and its
 * max-over-ranks clock without a second communication library. */
int smesh_comm_allreduce_f64(smesh_comm_t* c, double* values, int n, int op);

/* ---- annotation renderer: fused annotations gathered back to an image ------------------------- */
/* Replaces ModelAggregator::renderer() + ModelRenderer::render (include/semantic_meshes/fusion/Mesh.h:124-129,
 * 25-42; the harness does the same with tf.gather, eval-scannet/eval_scannet.py:314): the renderer holds
 * a snapshot of get() (float32[P,C]); render writes out[x,y,:] = snapshot[idx[x,y]] if idx < P else
 * background[:] into a contiguous float32 (W,H,C) image.  `background` is C floats in HOST memory. */
typedef struct smesh_annotation_renderer smesh_annotation_renderer_t;
int smesh_aggregator_renderer(smesh_aggregator_t* a, smesh_annotation_renderer_t** out);
int smesh_annotation_renderer_render(smesh_annotation_renderer_t* r,
                                     const void* indices, int idx_dtype, const int64_t idx_strides[2], int idx_memkind,
                                     const float* background, float* out, int out_memkind,
                                     uint64_t width, uint64_t height);
int smesh_annotation_renderer_destroy(smesh_annotation_renderer_t* r);

/* ---- fused view: render(camera) -> add(indices, probs) without leaving the device ----------- */
/* One iteration of the loop at python/scripts/colorize_cityscapes_mesh.py:54-67. probs/weights as
 * in smesh_aggregator_add with contiguous (W,H,C)/(W,H) layout. */
int smesh_fuse_view(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* camera,
                    const float* probs, const float* weights, int memkind);

/* A batch of `n` views: smesh_fuse_view(r, a, &cameras[i], probs[i], weights ? weights[i] : NULL, memkind) for i = 0 .. n-1, in
 * that order (the loop of python/scripts/colorize_cityscapes_mesh.py:54-67 handed over whole).  `probs` / `weights` are arrays of n
 * image pointers (weights, or single entries of it, may be NULL).  With a triangle renderer and device-resident images, up to eight
 * views share each rasteriser launch and each fusion launch (two for the Mul aggregator with 41 .. 48 classes): every accumulator row
 * makes ONE round trip for all of them, with the additions in the order of n separate calls. */
int smesh_fuse_views(smesh_renderer_t* r, smesh_aggregator_t* a, const smesh_camera_t* cameras, uint64_t n,
                     const float* const* probs, const float* const* weights, int memkind);

/* add() for an index image that is the UNMODIFIED device output `indices_dev` of one of r's six most recent
 * smesh_renderer_render_device() calls: the reference's two-call convention `idx, depth = renderer.render(cam);
 * aggregator.add(idx, probs)` (python/scripts/colorize_cityscapes_mesh.py:65-67) then runs the same triangle-order
 * fusion as smesh_fuse_view, using the per-triangle records that render left behind.  Same results as smesh_aggregator_add, but
 * ASYNCHRONOUS for DEVICE probs / weights like smesh_fuse_view: keep them valid until smesh_stream_release / smesh_synchronize;
 * if `idx_dev` is anything else, or the layout is not the dense (W,H[,C]) one, the call IS smesh_aggregator_add. */
int smesh_aggregator_add_rendered(smesh_aggregator_t* a, smesh_renderer_t* r, const uint32_t* idx_dev,
                                  const float* probs, const int64_t probs_strides[3], int probs_mem,
                                  const float* weights, const int64_t w_strides[2], int w_mem, uint64_t W, uint64_t H);

/* add() for an index image that has lost its identity but not its CONTENT: an image that went through another framework or the
 * host (DLPack -> TF -> numpy -> add, eval-scannet/eval_scannet.py:211-238).  If `indices` is a dense (W,H) uint32 / int32 image, HOST
 * or DEVICE, whose 64-bit content checksum (sum over the pixels of a bijective mix of position and value: a single changed pixel
 * always changes it) equals that of a plane produced by one of r's six most recent smesh_renderer_render_device() calls whose
 * per-triangle records are still held, the view is fused in triangle order exactly like smesh_aggregator_add_rendered -- reading
 * the given image -- and *matched = 1.  Otherwise NOTHING is added and *matched = 0: call smesh_aggregator_add.  Only planes that
 * were sealed (smesh_renderer_seal_render) take part.  Waits once for the checksum; the fusion itself is asynchronous for DEVICE
 * images like smesh_aggregator_add_rendered. */
int smesh_aggregator_add_matched(smesh_aggregator_t* a, smesh_renderer_t* r,
                                 const void* indices, int idx_dtype, const int64_t idx_strides[2], int idx_memkind,
                                 const float* probs, const int64_t probs_strides[3], int probs_mem,
                                 const float* weights, const int64_t w_strides[2], int w_mem,
                                 uint64_t width, uint64_t height, int* matched);

/* Computes (once) the content checksum of the plane `indices_dev` returned by one of r's last smesh_renderer_render_device() calls,
 * which is what makes copies of it recognisable by smesh_aggregator_add_matched.  Call it while the plane still holds what the
 * rasteriser wrote: before handing the pointer to another framework or copying the image to the host (the Python layer does, in
 * `__cuda_array_interface__` / `__dlpack__` / numpy conversion).  The plain render -> add loop neither needs nor pays it. */
int smesh_renderer_seal_render(smesh_renderer_t* r, const uint32_t* indices_dev);

/* Name of the fusion kernel the last smesh_fuse_view() / smesh_aggregator_add_rendered() on this thread dispatched ("k_fuse_tri": triangle-order
 * gather-accumulate, no atomics; "k_fuse_tri_any": the same for any class count; "k_scatter_strip": generic segmented scatter-add).  For reporting. */
const char* smesh_last_fuse_kernel(void);

/* Where the last add / fuse call on this thread got its per-primitive pixel lists from: "render-records" (left by the rasteriser:
 * smesh_fuse_view(s), smesh_aggregator_add_rendered / _add_matched), "image-records" (rebuilt from the index image itself:
 * smesh_aggregator_add on any dense image -- Mesh.h:90-106 on an image from anywhere, e.g. eval_scannet.py:168-185's .npz cache),
 * "scatter" (atomic scatter-add: what the triangle-order kernels do not take), "none".  For reporting. */
const char* smesh_last_add_path(void);

/* Diagnostics of the rasteriser's size classes (new, no reference counterpart; TriangleRenderer.h:92 leaves every triangle to one
 * DeviceMutexRasterizer).  `huge_stage_needed`: bit 0 clear if the library can PROVE from the mesh's bounding box and longest edge
 * that, seen from `camera`, no triangle crosses the near plane or has a screen box beyond 64 pixels a side -- it then leaves out the
 * launch that rasterises such triangles; bit 1 clear if it can prove that no box exceeds 8 x 8 pixels -- smesh_fuse_views then leaves
 * out the waves / launches that fuse such triangles; 3 without either proof.  `queue_lengths` (may be NULL): after r's last smesh_renderer_render /
 * _render_device (waits for it), [0] triangles with a box over 8 x 8 pixels, [1] nonzero if fragment queues overflowed, [2] of
 * [0], those beyond 64 pixels a side or clipped at the near plane, [3] of [0], those of at most 256 box pixels. */
int smesh_renderer_render_stats(smesh_renderer_t* r, const smesh_camera_t* camera, int* huge_stage_needed, uint32_t queue_lengths[4]);
/* The bound those proofs rest on, as plain host arithmetic (no device needed; tests/test_host.py checks it against brute force): an
 * upper bound on the screen extent, in pixels along either axis, of every triangle of the mesh that reaches `camera`'s image and has
 * no vertex at or behind the near plane -- +inf when nothing can be proven (a vertex may lie behind the camera, coarse triangles,
 * non-finite vertices).  <= 60: no triangle is huge or clipped; <= 6.5: no box exceeds 8 x 8 pixels. */
int smesh_box_extent_bound(const float* vertices, uint64_t num_vertices, const int32_t* faces, uint64_t num_faces,
                           const smesh_camera_t* camera, double* bound);

/* ---- timing hooks (tracing) -------------------------------------------- */
/* `slot_mask` is a bitmask of SMESH_PROF_* slots (bit s = slot s; 0 = off, 0xFF = all).
 * For every enabled slot the library brackets the kernels with HIP events on its own stream.
 * smesh_profile_read returns accumulated device milliseconds and launch counts per slot. */
#define SMESH_PROF_FUSE_SCATTER 0   /* the scatter-add fusion kernel          */
#define SMESH_PROF_FUSE_HIST    1   /* per-view histogram (Mesh.h:90-93)       */
#define SMESH_PROF_RASTER       2   /* all rasterizer kernels of one render    */
#define SMESH_PROF_FINALIZE     3   /* get() normalisation                     */
#define SMESH_PROF_EXCHANGE     4   /* the RCCL all-reduce / reduce-scatter    */
#define SMESH_PROF_SLOTS        8
int smesh_profile_enable(int device, int slot_mask);
/* Bracket only every n-th region of an enabled slot (default 1 = all): a HIP event pair costs ~4 us of stream
 * time, 4 % of a cfg2 view.  `launches` of smesh_profile_read counts the bracketed regions. */
int smesh_profile_sample_every(int device, uint32_t n);
int smesh_profile_read(int device, int slot, double* total_ms, uint64_t* launches);
/* The same, plus what the bracketed regions contained: `regions` = regions timed (== `launches` of smesh_profile_read),
 * `kernel_launches` = launches of the slot's dominant kernel inside them, `views` = views those launches processed (a two-view
 * fusion launch counts 1 and 2).  total_ms / kernel_launches is the kernel's average duration and views / kernel_launches the
 * views per launch, whatever the grouping of views into calls was.  Any out pointer may be NULL. */
int smesh_profile_read_ex(int device, int slot, double* total_ms, uint64_t* regions, uint64_t* kernel_launches, uint64_t* views);
/* Regions of the slot entered while it was enabled, bracketed or not.  A region is one kernel launch, except that smesh_fuse_views
 * makes ONE region of the back-to-back fusion launches of a group of views (an event pair around a single launch adds the dispatch
 * latency that back-to-back launches hide). */
int smesh_profile_regions(int device, int slot, uint64_t* entered);
int smesh_profile_reset(int device);

/* ---- synthetic workload generators (benchmark utility) ------------------------ */
/* Fills float32[N,C] class probabilities (softmax of hashed logits; `zero_fraction` of the pixels
 * get an all-zero don't-care row) directly in device (or host) memory. */
int smesh_synth_probs(float* out, uint64_t num_pixels, uint32_t num_classes, uint64_t seed,
                      float zero_fraction, int device, int memkind);
/* Raw device memory for benchmark inputs that must be resident in HBM before timing starts. */
int smesh_device_malloc(int device, uint64_t bytes, void** out);
int smesh_device_free(int device, void* ptr);
/* Device blocks released by the library's handles and by smesh_device_free stay mapped in a per-process cache and are handed out
 * again for requests of their size class: memory that hipMalloc has only just mapped is where, on a GPU shared by several
 * processes, kernel writes were seen to go missing (csrc/context.cpp, tools/alloc_churn_repro.hip).  smesh_device_trim unmaps what
 * the cache holds on `device` (-1: all devices) and reports how many bytes that was (`cached_bytes` may be NULL).
 * SMESH_ALLOC_CACHE_MB caps the cache (default: a quarter of the device's memory, at most 64 GiB; 0 = no cache). */
int smesh_device_trim(int device, uint64_t* cached_bytes);
/* Page-locked host memory (hipHostMalloc): HOST images passed to add() / fuse_view() from such a buffer cross PCIe by DMA. */
int smesh_host_malloc(uint64_t bytes, void** out);
int smesh_host_free(void* ptr);
int smesh_memcpy(void* dst, const void* src, uint64_t bytes, int dst_memkind, int src_memkind, int device);

#ifdef __cplusplus
}
#endif
#endif /* SMESH_H */
