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

# render() hands out planes that are rasterised on first USE (SMESH_LAZY_RENDER=0: at once, as until round 5).  The reference's loop is
# `idx, depth = renderer.render(cam); aggregator.add(idx, probs)` (python/scripts/colorize_cityscapes_mesh.py:54-67): when the index
# plane goes straight into add() nobody ever looks at it, and the aggregator can take (camera, probs) into a group of eight views
# that share their launches (smesh_fuse_views) -- the batch entry point's throughput behind the per-view API.
# Anything that looks at a plane -- np.asarray, .ptr, __cuda_array_interface__, DLPack, a capsule, add() with host or foreign images --
# rasterises it first; the content is the same either way.
LAZY_RENDER = os.environ.get("SMESH_LAZY_RENDER", "1") != "0"


class _PendingRender:
    """A render(camera) that has not been run yet; shared by its two planes."""

    def __init__(self, renderer, camera):
        import threading
        self.renderer = renderer
        self.pod = _lib.CameraPOD.from_buffer_copy(camera._pod)     # (the caller may change or drop its Camera)
        self.W, self.H = camera.resolution
        self.lock = threading.RLock()     # (re-entrant: a sibling plane may be dropped by the collector while run() holds it)
        self.done = False
        self.ptrs = [0, 0]            # index plane, depth plane
        self.dead = [False, False]    # the plane's array was dropped before the render ran

    def run(self):
        with self.lock:
            if self.done:
                return
            r = self.renderer
            pi, pd = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(_lib.lib().smesh_renderer_render_device(r._h, ctypes.byref(self.pod), ctypes.byref(pi), ctypes.byref(pd)))
            self.ptrs = [pi.value, pd.value]
            self.done = True
            r._lazy_unrun = False
            for which in (0, 1):
                if self.dead[which]:
                    self._release(which)

    def _release(self, which):
        h = self.renderer._h
        if h is not None and h.value and self.ptrs[which]:
            p = ctypes.c_void_p(self.ptrs[which])
            _lib.lib().smesh_renderer_release_image(h, p if which == 0 else None, p if which == 1 else None)
        self.ptrs[which] = 0

    def drop(self, which):
        with self.lock:
            if self.done:
                self._release(which)
            else:
                self.dead[which] = True


class _LazyPlane(DeviceArray):
    """One plane of a render() that runs on first use: reading `.ptr` (every consumer does) rasterises."""

    def __init__(self, pending, which, dtype, device, owner):
        self._pending = pending
        self._which = which
        DeviceArray.__init__(self, 0, (pending.W, pending.H), dtype, device, owner=owner)

    @property
    def ptr(self):
        pd = self._pending
        if not pd.done:
            pd.run()
        return pd.ptrs[self._which]

    @ptr.setter
    def ptr(self, value):      # (DeviceArray.__init__ assigns it)
        pass

    @property
    def unrun(self):
        return not self._pending.done

    def __del__(self):
        try:
            self._pending.drop(self._which)
        except Exception:
            pass



class _Renderer:
    """Common part of PlyRendererTriangles / PlyRendererTexels (Renderer.h:12-43)."""

    def __init__(self, handle, device, capsules=None):
        self._h = handle
        self.device = device
        self._primitives = None
        self._capsules = capsules     # what render() returns by default: None = this module's RETURN_CAPSULES (per renderer, not per process)
        self._lazy_unrun = False      # a lazy render() was handed out and has not been rasterised (yet): render_stats' queue lengths are older
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

    def render(self, camera, capsules=None, lazy=None):
        """Rasterise the mesh for `camera`; returns `(primitive_indices, depth)`:
        uint32 (W,H) with background 0xFFFFFFFF and float32 (W,H) with background +inf, device-resident.

        By default the two planes are `DeviceArray`s (`__dlpack__`, `__cuda_array_interface__`, `np.asarray`).  With
        `capsules=True` (or a renderer made by `semantic_meshes.render.triangles / texels` -- the reference's package name --, or
        SMESH_RENDER_CAPSULES=1) they are the `"dltensor"` PyCapsules the reference returns (Renderer.h:37-38), which
        `tf.experimental.dlpack.from_dlpack` (eval-scannet/eval_scannet.py:211-212) requires; `MeshAggregator.add` takes either."""
        if not isinstance(camera, Camera):
            raise TypeError("render() expects a semantic_meshes data.Camera")
        W, H = camera.resolution
        if capsules is None:
            capsules = RETURN_CAPSULES if self._capsules is None else self._capsules
        if lazy is None:
            lazy = LAZY_RENDER
        if lazy and not capsules:
            # (`lazy=False` / SMESH_LAZY_RENDER=0: rasterise now -- what a timing loop around render() alone wants)
            if W <= 0 or H <= 0 or W > 65536 or H > 65536:       # (what the library's check_camera refuses, refused now)
                raise ValueError("camera resolution must be in [1, 65536]")
            if W * H >= 0x7FFFFFFF // 4:
                raise ValueError("image too large")
            pend = _PendingRender(self, camera)
            indices = _LazyPlane(pend, 0, np.uint32, self.device, self)
            indices._rendered_by = self
            depth = _LazyPlane(pend, 1, np.float32, self.device, self)
            self._lazy_unrun = True
            return indices, depth
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
        self._lazy_unrun = False
        if capsules:
            return indices.capsule(), depth.capsule()
        return indices, depth

    def render_stats(self, camera, queues=True):
        """Diagnostics (`smesh_renderer_render_stats`): `(huge_stage_needed, queue_lengths)` -- whether the launch for triangles that
        cross the near plane or span more than 64 pixels is needed for `camera` (False: the library proved from the mesh's bounding
        box and longest edge that there is none), and after the last `render()` / `render_numpy()` the four queue lengths
        `[boxes over 8 x 8, queue overflow flag, of those huge or clipped, of those at most 256 box pixels]`.
        `self.last_big_stage_needed`: likewise for boxes over 8 x 8 pixels (what `fuse_views` leaves out where it is False)."""
        if queues and self._lazy_unrun:
            # the last render() has not been rasterised (its planes were not looked at, or went into a deferred add): the queue
            # lengths the library holds are an older render's -- rasterise this camera now
            _lib.flush_pending(self.device)
            self.render(camera, capsules=False, lazy=False)
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
