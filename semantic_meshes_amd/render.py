"""`semantic_meshes.render`: `triangles(mesh)` / `texels(mesh, cameras[, texels_per_pixel])`.

Reference: /root/reference/python/semantic_meshes/src/Render.cu:4-24 (module), include/Renderer.h:25-43
(`render(camera) -> (indices, depth)`), include/Ply.h:56-124 (factories).  The returned planes are
device-resident `DeviceArray`s in the reference's (W,H) y-fastest layout and can be passed straight
to `MeshAggregator.add` like the reference's DLPack capsules (python/scripts/colorize_cityscapes_mesh.py:65-67).
"""
import ctypes

import numpy as np

from . import _lib
from .data import Camera
from .device import DeviceArray


def _mesh_arrays(mesh):
    v = np.ascontiguousarray(getattr(mesh, "vertices"), dtype=np.float32)
    f = np.ascontiguousarray(getattr(mesh, "faces"), dtype=np.int32)
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
        raise ValueError("mesh must provide vertices[V,3] and faces[F,3]")
    return v, f


import os
import weakref

RETURN_CAPSULES = os.environ.get("SMESH_RENDER_CAPSULES", "0") == "1"   # render() returns "dltensor" PyCapsules like the reference

_live_renderers = weakref.WeakSet()   # MeshAggregator.add looks here for the render a foreign index image is a copy of


class _Renderer:
    """Common part of PlyRendererTriangles / PlyRendererTexels (Renderer.h:12-43)."""

    def __init__(self, handle, device, capsules=None):
        self._h = handle
        self.device = device
        self._primitives = None
        self._capsules = capsules     # what render() returns by default: None = this module's RETURN_CAPSULES (per renderer, not per process)
        _live_renderers.add(self)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                _lib.lib().smesh_renderer_destroy(h)
            except Exception:
                pass

    def getPrimitivesNum(self):
        if self._primitives is None:
            n = ctypes.c_uint64()
            _lib.check(_lib.lib().smesh_renderer_num_primitives(self._h, ctypes.byref(n)))
            self._primitives = int(n.value)
        return self._primitives

    def render(self, camera, capsules=None):
        """Rasterise the mesh for `camera`; returns `(primitive_indices, depth)`:
        uint32 (W,H) with background 0xFFFFFFFF and float32 (W,H) with background +inf, device-resident.

        By default the two planes are `DeviceArray`s (`__dlpack__`, `__cuda_array_interface__`, `np.asarray`).  With
        `capsules=True` (or a renderer made by `semantic_meshes.render.triangles / texels` -- the reference's package name --, or
        SMESH_RENDER_CAPSULES=1) they are the `"dltensor"` PyCapsules the reference returns (Renderer.h:37-38), which
        `tf.experimental.dlpack.from_dlpack` (eval-scannet/eval_scannet.py:211-212) requires; `MeshAggregator.add` takes either."""
        if not isinstance(camera, Camera):
            raise TypeError("render() expects a semantic_meshes data.Camera")
        W, H = camera.resolution
        pi, pd = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_renderer_render_device(self._h, ctypes.byref(camera._pod), ctypes.byref(pi), ctypes.byref(pd)))
        h = self._h

        def rel_i(ptr, h=h):
            if h.value:
                _lib.lib().smesh_renderer_release_image(h, ctypes.c_void_p(ptr), None)

        def rel_d(ptr, h=h):
            if h.value:
                _lib.lib().smesh_renderer_release_image(h, None, ctypes.c_void_p(ptr))

        indices = DeviceArray(pi.value, (W, H), np.uint32, self.device, owner=self, on_release=rel_i)
        indices._rendered_by = self   # add(indices, ...) can then reuse what this render left on the device
        depth = DeviceArray(pd.value, (W, H), np.float32, self.device, owner=self, on_release=rel_d)
        if capsules is None:
            capsules = RETURN_CAPSULES if self._capsules is None else self._capsules
        if capsules:
            return indices.capsule(), depth.capsule()
        return indices, depth

    def render_stats(self, camera, queues=True):
        """Diagnostics (`smesh_renderer_render_stats`): `(huge_stage_needed, queue_lengths)` -- whether the launch for triangles that
        cross the near plane or span more than 64 pixels is needed for `camera` (False: the library proved from the mesh's bounding
        box and longest edge that there is none), and after the last `render()` / `render_numpy()` the four queue lengths
        `[boxes over 8 x 8, queue overflow flag, of those huge or clipped, of those at most 256 box pixels]`.
        `self.last_big_stage_needed`: likewise for boxes over 8 x 8 pixels (what `fuse_views` leaves out where it is False)."""
        needed = ctypes.c_int()
        q = (ctypes.c_uint32 * 4)()
        _lib.check(_lib.lib().smesh_renderer_render_stats(self._h, ctypes.byref(camera._pod), ctypes.byref(needed),
                                                         ctypes.cast(q, ctypes.c_void_p) if queues else None))
        self.last_big_stage_needed = bool(needed.value & 2)
        return bool(needed.value & 1), [int(x) for x in q]

    def render_numpy(self, camera):
        """Host variant: one call, results copied into fresh numpy arrays."""
        W, H = camera.resolution
        idx = np.empty((W, H), np.uint32)
        depth = np.empty((W, H), np.float32)
        _lib.check(_lib.lib().smesh_renderer_render(self._h, ctypes.byref(camera._pod),
                                                   idx.ctypes.data_as(ctypes.c_void_p), depth.ctypes.data_as(ctypes.c_void_p)))
        return idx, depth


class PlyRendererTriangles(_Renderer):
    """Triangle primitives: id == ordinal of the face in the mesh (TriangleRenderer.h:41-44,57-60)."""

    def __init__(self, mesh, device=0, capsules=None):
        v, f = _mesh_arrays(mesh)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_renderer_create_triangles(v.ctypes.data_as(ctypes.c_void_p), len(v),
                                                             f.ctypes.data_as(ctypes.c_void_p), len(f), device, ctypes.byref(h)))
        super().__init__(h, device, capsules)


class PlyRendererTexels(_Renderer):
    """Texel primitives (TexturedTriangleRenderer.h:87-182)."""
    is_texel = True      # (MeshAggregator.add: copies of a texel render keep going through the content match)

    def __init__(self, mesh, cameras, texels_per_pixel=0.1, device=0, capsules=None):
        v, f = _mesh_arrays(mesh)
        cams = list(cameras.getCameras()) if hasattr(cameras, "getCameras") else list(cameras)
        for c in cams:
            if not isinstance(c, Camera):
                raise TypeError("texels() expects a list of data.Camera")
        pods = (_lib.CameraPOD * max(len(cams), 1))(*[c._pod for c in cams])
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_renderer_create_texels(v.ctypes.data_as(ctypes.c_void_p), len(v),
                                                          f.ctypes.data_as(ctypes.c_void_p), len(f),
                                                          ctypes.cast(pods, ctypes.c_void_p), len(cams),
                                                          float(texels_per_pixel), device, ctypes.byref(h)))
        super().__init__(h, device, capsules)
        self._num_faces = len(f)

    def texel_layout(self):
        """(faces after the reference's vertex re-ordering, per-triangle resolution, first texel id)."""
        F = self._num_faces
        faces, res, first = np.empty((F, 3), np.int32), np.empty(F, np.uint32), np.empty(F, np.uint32)
        _lib.check(_lib.lib().smesh_renderer_texel_layout(self._h, faces.ctypes.data_as(ctypes.c_void_p),
                                                         res.ctypes.data_as(ctypes.c_void_p), first.ctypes.data_as(ctypes.c_void_p)))
        return faces, res, first


def triangles(mesh, device=0, capsules=None):
    """`semantic_meshes.render.triangles(mesh)` (Render.cu:24).  `capsules`: what the renderer's `render()` returns by default
    (None: DeviceArrays unless SMESH_RENDER_CAPSULES=1; True: the reference's "dltensor" PyCapsules)."""
    return PlyRendererTriangles(mesh, device=device, capsules=capsules)


def texels(mesh, cameras, texels_per_pixel=0.1, device=0, capsules=None):
    """`semantic_meshes.render.texels(mesh, colmap|[cameras][, texels_per_pixel])` (Render.cu:20-23);
    default texels_per_pixel 0.1 (TexturedTriangleRenderer.h:87)."""
    return PlyRendererTexels(mesh, cameras, texels_per_pixel, device=device, capsules=capsules)
