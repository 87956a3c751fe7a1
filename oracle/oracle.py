"""ctypes binding of the CPU oracle (oracle/libsmesh_oracle.so).

*** TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT. ***
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Nothing under semantic_meshes_amd/ imports it, and it never touches the HIP library.

The classes take and return plain numpy arrays in the reference's (W,H[,C]) layout
(/root/reference/python/semantic_meshes/include/Renderer.h:29,32).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SMESH_ORACLE_LIB: another build of the same source (the sanitizer builds of oracle/Makefile, tools/oracle_sanitizers.sh)
_LIB_PATH = os.environ.get("SMESH_ORACLE_LIB") or os.path.join(_HERE, "libsmesh_oracle.so")

AGG_KINDS = {"sum": 0, "summax": 1, "mul": 2}
_IDX_DTYPES = {np.dtype(np.uint32): 0, np.dtype(np.int32): 1, np.dtype(np.uint64): 2, np.dtype(np.int64): 3}


class CameraPOD(ctypes.Structure):
    # mirrors smesh_camera_t in include/smesh.h
    _fields_ = [
        ("rotation", ctypes.c_float * 9),
        ("translation", ctypes.c_float * 3),
        ("focal", ctypes.c_double * 2),
        ("principal", ctypes.c_double * 2),
        ("width", ctypes.c_uint64),
        ("height", ctypes.c_uint64),
    ]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    if os.environ.get("SMESH_ORACLE_LIB"):
        return _LIB_PATH
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "smesh_oracle.cpp"))
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libsmesh_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.smesh_last_error.restype = ctypes.c_char_p
        L.smesh_backend.restype = ctypes.c_char_p
        _lib = L
    return _lib


def _check(status):
    if status != 0:
        msg = lib().smesh_last_error().decode()
        raise (ValueError if status == 1 else RuntimeError)(msg)


def make_camera(rotation, translation, resolution, focal_lengths, principal_point):
    """Same conversions as the reference ctor (/root/reference/python/semantic_meshes/include/Camera.h:16-57)."""
    cam = CameraPOD()
    R = np.asarray(rotation, dtype=np.float32).reshape(3, 3)
    t = np.asarray(translation, dtype=np.float32).reshape(3)
    f = np.asarray(focal_lengths, dtype=np.float32).reshape(2).astype(np.float64)
    c = np.asarray(principal_point, dtype=np.float32).reshape(2).astype(np.float64)
    cam.rotation[:] = R.reshape(-1).tolist()
    cam.translation[:] = t.tolist()
    cam.focal[:] = f.tolist()
    cam.principal[:] = c.tolist()
    cam.width = int(resolution[0])
    cam.height = int(resolution[1])
    return cam


def _as_pod(camera):
    if isinstance(camera, CameraPOD):
        return camera
    # duck-typed product Camera (semantic_meshes_amd.data.Camera): copy its fields
    return make_camera(camera.rotation, camera.translation, camera.resolution, camera.focal_lengths, camera.principal_point)


def set_threads(n):
    lib().smesh_oracle_set_threads(int(n))


def get_threads():
    return int(lib().smesh_oracle_get_threads())


def set_accum_double(on):
    """float64 accumulators (tolerance reference); the faithful restatement uses float32."""
    lib().smesh_oracle_set_accum_double(1 if on else 0)


def set_fast_histogram(on):
    """"Optimised CPU" baseline variant: dense parallel histogram instead of the reference's serial std::map (same counts)."""
    lib().smesh_oracle_set_fast_histogram(1 if on else 0)


def set_mul_literal(on):
    """Independent yardstick: the Mul term as the literal reading of Fusion.cu:83-87, logf(powf(p, w)) with p^w rounded to
    float32 first, instead of the spec'd w * log_spec(p)."""
    lib().smesh_oracle_set_mul_literal(1 if on else 0)


def set_edge_five_op(on):
    """Independent yardstick: the round-1 edge function sign * (dx (py - ly) - dy (px - lx)) instead of the spec'd
    fma(A, px, fma(B, py, C))."""
    lib().smesh_oracle_set_edge_five_op(1 if on else 0)


class OracleRenderer:
    def __init__(self, vertices, faces, cameras=None, texels_per_pixel=0.1):
        self.vertices = np.ascontiguousarray(vertices, dtype=np.float32).reshape(-1, 3)
        self.faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1, 3)
        self._h = ctypes.c_void_p()
        vp = self.vertices.ctypes.data_as(ctypes.c_void_p)
        fp = self.faces.ctypes.data_as(ctypes.c_void_p)
        if cameras is None:
            _check(lib().smesh_renderer_create_triangles(vp, ctypes.c_uint64(len(self.vertices)), fp,
                                                         ctypes.c_uint64(len(self.faces)), 0, ctypes.byref(self._h)))
        else:
            pods = (CameraPOD * len(cameras))(*[_as_pod(c) for c in cameras])
            _check(lib().smesh_renderer_create_texels(vp, ctypes.c_uint64(len(self.vertices)), fp,
                                                      ctypes.c_uint64(len(self.faces)), pods, ctypes.c_uint64(len(cameras)),
                                                      ctypes.c_float(texels_per_pixel), 0, ctypes.byref(self._h)))
        self.is_texels = cameras is not None

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().smesh_renderer_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:  # interpreter shutdown
            pass

    def getPrimitivesNum(self):
        n = ctypes.c_uint64()
        _check(lib().smesh_renderer_num_primitives(self._h, ctypes.byref(n)))
        return int(n.value)

    def texel_layout(self):
        F = len(self.faces)
        faces = np.empty((F, 3), np.int32)
        res = np.empty(F, np.uint32)
        first = np.empty(F, np.uint32)
        _check(lib().smesh_renderer_texel_layout(self._h, faces.ctypes.data_as(ctypes.c_void_p),
                                                 res.ctypes.data_as(ctypes.c_void_p), first.ctypes.data_as(ctypes.c_void_p)))
        return faces, res, first

    def render(self, camera):
        cam = _as_pod(camera)
        W, H = int(cam.width), int(cam.height)
        idx = np.empty((W, H), np.uint32)
        depth = np.empty((W, H), np.float32)
        _check(lib().smesh_renderer_render(self._h, ctypes.byref(cam), idx.ctypes.data_as(ctypes.c_void_p),
                                           depth.ctypes.data_as(ctypes.c_void_p)))
        return idx, depth


def _strides(arr):
    return (ctypes.c_int64 * arr.ndim)(*[s // arr.itemsize for s in arr.strides])


class OracleAggregator:
    def __init__(self, primitives, classes, aggregator="sum", images_equal_weight=0.5):
        self.P, self.C = int(primitives), int(classes)
        self.primitives, self.classes = self.P, self.C     # (the product aggregator's attribute names)
        self.kind = aggregator[:1].upper() + aggregator[1:].lower()
        self._h = ctypes.c_void_p()
        _check(lib().smesh_aggregator_create(ctypes.c_uint64(self.P), ctypes.c_uint32(self.C), AGG_KINDS[aggregator.lower()],
                                             ctypes.c_float(images_equal_weight), 0, ctypes.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().smesh_aggregator_destroy(self._h)
                self._h = ctypes.c_void_p()
        except Exception:  # interpreter shutdown
            pass

    def add(self, idx, probs, weights=None):
        idx = np.asarray(idx)
        probs = np.asarray(probs, dtype=np.float32)
        if idx.dtype not in _IDX_DTYPES:
            raise ValueError("unsupported index dtype %s" % idx.dtype)
        if idx.ndim != 2 or probs.ndim != 3 or idx.shape != probs.shape[:2] or probs.shape[2] != self.C:
            raise ValueError("shape mismatch")
        if any(s < 0 for s in idx.strides + probs.strides):
            idx, probs = np.ascontiguousarray(idx), np.ascontiguousarray(probs)
        wp, ws = None, None
        if weights is not None:
            weights = np.asarray(weights, dtype=np.float32)
            if weights.shape != idx.shape:
                raise ValueError("shape mismatch")
            if any(s < 0 for s in weights.strides):
                weights = np.ascontiguousarray(weights)
            wp, ws = weights.ctypes.data_as(ctypes.c_void_p), _strides(weights)
        W, H = idx.shape
        _check(lib().smesh_aggregator_add(self._h, idx.ctypes.data_as(ctypes.c_void_p), _IDX_DTYPES[idx.dtype], _strides(idx), 0,
                                          probs.ctypes.data_as(ctypes.c_void_p), _strides(probs), 0, wp, ws, 0,
                                          ctypes.c_uint64(W), ctypes.c_uint64(H)))

    def reset(self):
        _check(lib().smesh_aggregator_reset(self._h))

    def get(self):
        out = np.empty((self.P, self.C), np.float32)
        _check(lib().smesh_aggregator_get(self._h, out.ctypes.data_as(ctypes.c_void_p), 0))
        return out

    def get_rows(self, row_lo, row_hi):
        out = np.empty((int(row_hi) - int(row_lo), self.C), np.float32)
        _check(lib().smesh_aggregator_get_rows(self._h, ctypes.c_uint64(int(row_lo)), ctypes.c_uint64(int(row_hi)),
                                               out.ctypes.data_as(ctypes.c_void_p), 0))
        return out

    def get_raw(self):
        out = np.empty((self.P, self.C), np.float32)
        _check(lib().smesh_aggregator_get_raw(self._h, out.ctypes.data_as(ctypes.c_void_p), 0))
        return out

    def set_raw(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        if raw.shape != (self.P, self.C):
            raise ValueError("shape mismatch")
        _check(lib().smesh_aggregator_set_raw(self._h, raw.ctypes.data_as(ctypes.c_void_p), 0))


    def get_raw_rows(self, row_lo, row_hi, plane=0):
        out = np.empty((int(row_hi) - int(row_lo), self.C), np.float32)
        _check(lib().smesh_aggregator_get_raw_rows(self._h, ctypes.c_uint64(int(row_lo)), ctypes.c_uint64(int(row_hi)), int(plane),
                                                   out.ctypes.data_as(ctypes.c_void_p), 0))
        return out

    def set_raw_rows(self, row_lo, raw, plane=0):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        _check(lib().smesh_aggregator_set_raw_rows(self._h, ctypes.c_uint64(int(row_lo)), ctypes.c_uint64(int(row_lo) + raw.shape[0]), int(plane),
                                                   raw.ctypes.data_as(ctypes.c_void_p), 0))


def render_annotations(aggregator, idx, background):
    """ModelAggregator::renderer() + ModelRenderer::render (Mesh.h:25-42,124-129) on the oracle."""
    idx = np.ascontiguousarray(idx)
    bg = np.ascontiguousarray(background, dtype=np.float32)
    h = ctypes.c_void_p()
    _check(lib().smesh_aggregator_renderer(aggregator._h, ctypes.byref(h)))
    W, H = idx.shape
    out = np.empty((W, H, aggregator.C), np.float32)
    try:
        _check(lib().smesh_annotation_renderer_render(h, idx.ctypes.data_as(ctypes.c_void_p), _IDX_DTYPES[idx.dtype], _strides(idx), 0,
                                                      bg.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), 0,
                                                      ctypes.c_uint64(W), ctypes.c_uint64(H)))
    finally:
        lib().smesh_annotation_renderer_destroy(h)
    return out


def synth_probs(num_pixels, classes, seed, zero_fraction=0.0):
    out = np.empty((int(num_pixels), int(classes)), np.float32)
    _check(lib().smesh_synth_probs(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(int(num_pixels)), ctypes.c_uint32(int(classes)),
                                   ctypes.c_uint64(int(seed) & (2**64 - 1)), ctypes.c_float(zero_fraction), 0, 0))
    return out
