"""`semantic_meshes.fusion.MeshAggregator`.

Reference: /root/reference/python/semantic_meshes/src/Fusion.cu:120-150 (factory, defaults "sum", 0.5),
include/Fusion.h:42-76 (add1/add2/reset/get), /root/reference/include/semantic_meshes/fusion/Mesh.h:65-132.
The class count is a run-time value here (compile-time CLASSES_NUMS list in the reference).
"""
import ctypes
import threading

import numpy as np

from . import _lib
import os

from .device import DeviceArray, describe, release_to, result_empty

_IDX_CODES = {np.dtype(np.uint32): _lib.IDX_U32, np.dtype(np.int32): _lib.IDX_I32,
              np.dtype(np.uint64): _lib.IDX_U64, np.dtype(np.int64): _lib.IDX_I64}


def _c64(vals):
    return (ctypes.c_int64 * len(vals))(*vals)


GROUP_VIEWS = 8                                                     # views per deferred group (= the library's views per launch)
DEFER_VIEWS = os.environ.get("SMESH_DEFER_VIEWS", "1") != "0"       # default of MeshAggregator.defer

import weakref                                                       # noqa: E402
_aggregators = weakref.WeakSet()
_aggregators_lock = threading.Lock()      # (aggregators are created and flushed from different threads in the harness)


def _flush_device(device=None):
    with _aggregators_lock:
        live = list(_aggregators)
    for a in live:
        if (device is None or a.device == device) and a._pending:
            a.flush()


_lib._flush_hooks.append(_flush_device)


class _MeshAggregator:
    def __init__(self, primitives, classes, kind, images_equal_weight, device):
        self.primitives, self.classes = int(primitives), int(classes)
        self.kind, self.images_equal_weight, self.device = kind, float(images_equal_weight), int(device)
        if self.primitives < 0 or self.classes <= 0:
            raise ValueError("primitives must be >= 0 and classes > 0")
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_aggregator_create(self.primitives, self.classes, _lib.AGG_KINDS[kind],
                                                     self.images_equal_weight, self.device, ctypes.byref(h)))
        self._handle = h
        self._inflight = []     # (completion token, [objects]) of asynchronous calls whose device inputs may still be being read
        self._inflight_lock = threading.Lock()   # (the harness adds from a worker thread while the main thread may call get())
        # Deferred views (round 6): add() of a not-yet-rasterised render() plane and fuse_view(), with class vectors in this library's own
        # device arrays, are taken into a group of up to eight views that go to the library as ONE smesh_fuse_views call -- the views
        # share their rasteriser launches and each accumulator row makes one round trip for all of them: the batch entry point's
        # throughput behind the reference's per-view loop (colorize_cityscapes_mesh.py:54-67).  Same sums in the same order.  The group
        # is handed over on the eighth view, at anything else that uses the aggregator (`_h`), at `flush()`, at `_lib.synchronize()`.
        self._pending = []      # [(renderer, CameraPOD, W, H, probs array, weights array or None)]
        self._pending_lock = threading.RLock()
        self.defer = DEFER_VIEWS
        with _aggregators_lock:
            _aggregators.add(self)

    @property
    def _h(self):
        """The library handle -- for a call that is about to use it: every deferred view is handed over first."""
        if self._pending:
            self.flush()
        return self._handle

    def flush(self):
        """Hand the deferred views (see __init__) to the library now.  Asynchronous like fuse_views: nothing is waited for."""
        with self._pending_lock:
            todo, self._pending = self._pending, []
            if not todo:
                return
            renderer = todo[0][0]
            n = len(todo)
            pods = (_lib.CameraPOD * n)(*[t[1] for t in todo])
            pptr = (ctypes.c_void_p * n)(*[t[4].ptr for t in todo])
            wptr = None
            if todo[0][5] is not None:
                wptr = (ctypes.c_void_p * n)(*[t[5].ptr for t in todo])
            _lib.check(_lib.lib().smesh_fuse_views(renderer._h, self._handle, pods, n, pptr, wptr, _lib.MEM_DEVICE))
            # (the class vectors are this library's own arrays: freed behind its streams, no completion token needed)

    def _deferrable(self, probs_image, weights_image, W, H):
        """May a view with these inputs wait for its group?  Only with inputs nobody else can write to behind our back: this library's
        own dense float32 device arrays that were never exported to another framework (anything else is consumed by the call itself,
        in order: foreign device tensors may be re-used by their owner as soon as the call returns, host arrays likewise)."""
        if not self.defer:
            return False
        ok = (type(probs_image) is DeviceArray and not probs_image._exported and probs_image.device == self.device
              and probs_image.dtype == np.float32 and probs_image.shape == (W, H, self.classes)
              and probs_image.strides == (H * self.classes, self.classes, 1))
        if ok and weights_image is not None:
            ok = (type(weights_image) is DeviceArray and not weights_image._exported and weights_image.device == self.device
                  and weights_image.dtype == np.float32 and weights_image.shape == (W, H) and weights_image.strides == (H, 1))
        return ok

    def _defer(self, renderer, pod, W, H, probs_image, weights_image):
        with self._pending_lock:
            if self._pending:
                first = self._pending[0]
                # one group = one renderer, one image size, weights for all views or for none
                if first[0] is not renderer or (first[2], first[3]) != (W, H) or (first[5] is None) != (weights_image is None):
                    self.flush()
            self._pending.append((renderer, pod, W, H, probs_image, weights_image))
            if len(self._pending) >= GROUP_VIEWS:
                self.flush()

    def _hold(self, keepalives):
        """The asynchronous entry points read DEVICE images after they return.  `release_to()` orders the stream `describe()` guessed for
        their owner behind those reads -- but a tensor that is later freed or re-used on ANOTHER stream or thread (or whose producer
        exported no stream) could be recycled by its caching allocator while the kernels still read it.  So the aggregator keeps its own
        references to the inputs of other frameworks until a completion token recorded behind the call is done (checked, without waiting,
        at the next call).  This library's own DeviceArrays are freed behind its streams and need none."""
        self._drain()
        foreign = [k for k in keepalives if k is not None and not isinstance(k, DeviceArray)]
        if not foreign:
            return
        tok = ctypes.c_uint64(0)
        _lib.check(_lib.lib().smesh_token_record(self.device, ctypes.byref(tok)))
        with self._inflight_lock:
            self._inflight.append((tok.value, foreign))

    def _drain(self):
        done = ctypes.c_int(0)
        with self._inflight_lock:      # (a token goes back to the library's pool exactly once)
            while self._inflight:
                _lib.check(_lib.lib().smesh_token_done(self.device, ctypes.c_uint64(self._inflight[0][0]), ctypes.byref(done)))
                if not done.value:
                    break
                self._inflight.pop(0)

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        self._pending = []      # (views nobody can ask the result of any more)
        if h is not None and h.value:
            try:
                _lib.lib().smesh_aggregator_destroy(h)      # (waits for the device: every outstanding token is done afterwards)
                done = ctypes.c_int(0)
                for tok, _ in getattr(self, "_inflight", []):
                    _lib.lib().smesh_token_done(self.device, ctypes.c_uint64(tok), ctypes.byref(done))   # hands the event back to the pool
                self._inflight = []
            except Exception:
                pass

    def add(self, primitive_image, probs_image, weights_image=None):
        """Fuse one view: `primitive_image` (W,H) of uint32/int32/uint64/int64, `probs_image` (W,H,C) float32,
        optional `weights_image` (W,H) float32; host numpy or device arrays, any non-negative strides."""
        if type(primitive_image).__name__ == "PyCapsule":
            # render() in capsule mode (the reference's return type) handed straight back, as python/scripts/colorize_cityscapes_mesh.py:65-67 does
            from . import dlpack
            own = dlpack.own_capsule_owner(primitive_image)
            if own is not None:
                primitive_image = own
        if (getattr(primitive_image, "unrun", False) and primitive_image._which == 0 and primitive_image.device == self.device
                and self._deferrable(probs_image, weights_image, *primitive_image.shape)):
            # render()'s index plane handed straight back, not rasterised yet (render.py: _LazyPlane): nobody has looked at it, so
            # (camera, probs) joins the aggregator's group of deferred views and the plane is never produced
            pend = primitive_image._pending
            if pend.W and pend.H:
                self._defer(pend.renderer, pend.pod, pend.W, pend.H, probs_image, weights_image)
            return
        streams = []   # streams of other frameworks whose device arrays this call reads (ordered before and after, no host wait)
        ip, imem, ishape, idt, istr, k0 = describe(primitive_image, 2, "primitive image", self.device, streams)
        pp, pmem, pshape, pdt, pstr, k1 = describe(probs_image, 3, "probs image", self.device, streams)
        if idt not in _IDX_CODES:
            raise ValueError("primitive image dtype must be one of uint32/int32/uint64/int64, got %s" % idt)
        if pdt != np.float32:
            if pmem == _lib.MEM_HOST and pdt.kind == "f":
                probs_image = np.asarray(probs_image, dtype=np.float32)
                pp, pmem, pshape, pdt, pstr, k1 = describe(probs_image, 3, "probs image", self.device, streams)
            else:
                raise ValueError("probs image must be float32, got %s" % pdt)
        wp, wmem, wstr, k2, wshape = None, _lib.MEM_HOST, None, None, None
        if weights_image is not None:
            wp, wmem, wshape, wdt, wstr, k2 = describe(weights_image, 2, "weights image", self.device, streams)
            if wdt != np.float32:
                if wmem == _lib.MEM_HOST and wdt.kind == "f":
                    weights_image = np.asarray(weights_image, dtype=np.float32)
                    wp, wmem, wshape, wdt, wstr, k2 = describe(weights_image, 2, "weights image", self.device, streams)
                else:
                    raise ValueError("weights image must be float32, got %s" % wdt)
        if tuple(ishape) != tuple(pshape[:2]) or (wshape is not None and tuple(wshape) != tuple(ishape)):
            # Mesh.h:68-74 std::invalid_argument
            raise ValueError("Primitive image %s, probs image %s and weights image %s must have the same width and height"
                             % (tuple(ishape), tuple(pshape[:2]), None if wshape is None else tuple(wshape)))
        if pshape[2] != self.classes:
            raise ValueError("probs image has %d classes, aggregator was built for %d" % (pshape[2], self.classes))
        W, H = ishape
        if W == 0 or H == 0:
            return
        rb = getattr(primitive_image, "_rendered_by", None)
        if (rb is not None and not primitive_image._exported and getattr(rb, "_h", None) is not None and rb._h.value
                and idt == np.uint32 and tuple(istr) == (H, 1)):
            # the untouched output of renderer.render(): the reference's two-call loop (colorize_cityscapes_mesh.py:65-67)
            # runs the same triangle-order fusion as fuse_view (the library re-checks that it is the latest render)
            _lib.check(_lib.lib().smesh_aggregator_add_rendered(
                self._h, rb._h, ctypes.c_void_p(ip), ctypes.c_void_p(pp), _c64(pstr), pmem,
                None if wp is None else ctypes.c_void_p(wp), None if wstr is None else _c64(wstr), wmem, W, H))
            release_to(self.device, streams)
            self._hold([k1 if pmem == _lib.MEM_DEVICE else None, k2 if (wp is not None and wmem == _lib.MEM_DEVICE) else None])
            return
        if idt.itemsize == 4 and tuple(istr) == (H, 1) and self.match_renders and not self._records_from_image():
            # An index image that went through another framework or numpy (DLPack -> TF -> .numpy() -> add in the reference's
            # harness, eval-scannet/eval_scannet.py:211-238): if its content checksum still equals that of one of the last renders of
            # a renderer with this many primitives on this GPU, the triangle-order fusion applies (smesh_aggregator_add_matched)
            from .render import _live_renderers
            for rb in list(_live_renderers):
                if rb.device != self.device or rb._h is None or not rb._h.value or rb.getPrimitivesNum() != self.primitives:
                    continue
                matched = ctypes.c_int(0)
                _lib.check(_lib.lib().smesh_aggregator_add_matched(
                    self._h, rb._h, ctypes.c_void_p(ip), _IDX_CODES[idt], _c64(istr), imem, ctypes.c_void_p(pp), _c64(pstr), pmem,
                    None if wp is None else ctypes.c_void_p(wp), None if wstr is None else _c64(wstr), wmem, W, H, ctypes.byref(matched)))
                if matched.value:
                    release_to(self.device, streams)
                    self._hold([k0 if imem == _lib.MEM_DEVICE else None, k1 if pmem == _lib.MEM_DEVICE else None,
                                k2 if (wp is not None and wmem == _lib.MEM_DEVICE) else None])
                    return
        # Host images are consumed before the call returns (Fusion.h:45-47).  Device images are read asynchronously: this library's own
        # DeviceArrays are freed behind its streams, other frameworks' streams are put behind the reads by release_to() -- so the
        # record passes of the next add() on a foreign image can run beside this call's fusion (fusion.hip, add_device).
        _lib.check(_lib.lib().smesh_aggregator_add_async(
            self._h, ctypes.c_void_p(ip), _IDX_CODES[idt], _c64(istr), imem,
            ctypes.c_void_p(pp), _c64(pstr), pmem,
            None if wp is None else ctypes.c_void_p(wp), None if wstr is None else _c64(wstr), wmem, W, H))
        release_to(self.device, streams)
        self._hold([k0 if imem == _lib.MEM_DEVICE else None, k1 if pmem == _lib.MEM_DEVICE else None,
                    k2 if (wp is not None and wmem == _lib.MEM_DEVICE) else None])

    def add_many(self, primitive_images, probs_images, weights_images=None):
        """`add()` for a batch of views, in order (new functionality; the reference's loop adds one image per call).  Same sums as
        the calls one by one -- per accumulator row the same float32 additions in the same order -- but device-resident dense
        uint32 / int32 index images with dense float32 device class vectors, all of one size, share their kernel launches in groups
        of up to eight (`smesh_aggregator_add_many`).  Anything else in the batch is added image by image."""
        prims, probs = list(primitive_images), list(probs_images)
        wts = None if weights_images is None else list(weights_images)
        n = len(prims)
        if len(probs) != n or (wts is not None and len(wts) != n):
            raise ValueError("add_many needs one probs image (and one weights image) per primitive image")
        if n == 0:
            return
        streams, desc = [], []
        for i in range(n):
            pi = prims[i]
            if type(pi).__name__ == "PyCapsule":
                from . import dlpack
                own = dlpack.own_capsule_owner(pi)
                if own is not None:
                    pi = own
            d_i = describe(pi, 2, "primitive image", self.device, streams)
            d_p = describe(probs[i], 3, "probs image", self.device, streams)
            d_w = None if wts is None or wts[i] is None else describe(wts[i], 2, "weights image", self.device, streams)
            desc.append((d_i, d_p, d_w))
        (ip0, imem0, ishape0, idt0, istr0, _), (pp0, pmem0, pshape0, pdt0, pstr0, _), w0 = desc[0]
        uniform = idt0 in _IDX_CODES and pdt0 == np.float32 and imem0 == _lib.MEM_DEVICE and pmem0 == _lib.MEM_DEVICE
        for d_i, d_p, d_w in desc:
            uniform = (uniform and d_i[1:5] == (imem0, ishape0, idt0, istr0) and d_p[1:5] == (pmem0, pshape0, pdt0, pstr0)
                       and (d_w is None) == (w0 is None)
                       and (d_w is None or (d_w[1] == _lib.MEM_DEVICE and d_w[3] == np.float32 and tuple(d_w[2]) == tuple(ishape0) and d_w[4] == w0[4])))
        if uniform and (tuple(ishape0) != tuple(pshape0[:2]) or pshape0[2] != self.classes):
            uniform = False      # (add() raises the reference's error for the image concerned)
        if not uniform or n < 2:
            for i in range(n):
                self.add(prims[i], probs[i], None if wts is None else wts[i])
            return
        W, H = ishape0
        if W == 0 or H == 0:
            return
        iptr = (ctypes.c_void_p * n)(*[d[0][0] for d in desc])
        pptr = (ctypes.c_void_p * n)(*[d[1][0] for d in desc])
        wptr = None if w0 is None else (ctypes.c_void_p * n)(*[d[2][0] for d in desc])
        _lib.check(_lib.lib().smesh_aggregator_add_many(
            self._h, n, iptr, _IDX_CODES[idt0], _c64(istr0), imem0, pptr, _c64(pstr0), pmem0,
            wptr, None if w0 is None else _c64(w0[4]), _lib.MEM_DEVICE if w0 is not None else _lib.MEM_HOST, W, H))
        release_to(self.device, streams)
        self._hold([d[0][5] for d in desc] + [d[1][5] for d in desc] + [d[2][5] for d in desc if d[2] is not None])

    # class-wide switch for the content check above
    match_renders = os.environ.get("SMESH_MATCH_RENDERS", "1") != "0"

    def _records_from_image(self):
        """smesh_aggregator_add rebuilds the per-primitive records from ANY dense image (image_records.hip; every class count since
        round 3, SMESH_ADD_RECORDS_MIN_C moves the threshold) and
        runs the same triangle-order kernels as a matched render would -- without the content checksum's read-back, which costs more
        than the records do.  (Texel renderers keep the match: their records carry the texel tables.)"""
        if os.environ.get("SMESH_ADD_RECORDS") == "0" or os.environ.get("SMESH_FUSE") == "strip":
            return False
        from .render import _live_renderers
        if any(getattr(rb, "is_texel", False) and rb.getPrimitivesNum() == self.primitives for rb in list(_live_renderers)):
            return False
        return self.classes >= int(os.environ.get("SMESH_ADD_RECORDS_MIN_C", "0"))

    def reset(self):
        with self._pending_lock:
            self._pending = []          # (deferred views whose sums would be cleared anyway)
            _lib.check(_lib.lib().smesh_aggregator_reset(self._handle))
        self._drain()

    def get(self):
        """Normalised per-primitive class distribution, fresh float32[P,C] numpy array (Fusion.h:72-76)."""
        out = result_empty((self.primitives, self.classes), np.float32)
        if out.size:
            _lib.check(_lib.lib().smesh_aggregator_get(self._h, out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))
        self._drain()     # (get() waited for the stream: every earlier call's reads are over)
        return out

    def get_rows(self, row_lo, row_hi):
        """`get()` for the rows [row_lo, row_hi) only (row_lo a multiple of 4): what a rank owns after
        `Communicator.reduce_scatter` (new functionality)."""
        row_lo, row_hi = int(row_lo), int(row_hi)
        out = result_empty((max(row_hi - row_lo, 0), self.classes), np.float32)
        _lib.check(_lib.lib().smesh_aggregator_get_rows(self._h, row_lo, row_hi, out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))
        return out

    def get_device(self):
        """`get()` without the trip to the host: the normalised float32[P,C] result as a device-resident `DeviceArray`
        (`__cuda_array_interface__` / DLPack) in a fresh HBM allocation owned by the returned object."""
        from .device import DeviceBuffer
        buf = DeviceBuffer(max(self.primitives * self.classes * 4, 4), self.device)
        if self.primitives * self.classes:
            _lib.check(_lib.lib().smesh_aggregator_get(self._h, ctypes.c_void_p(buf.ptr), _lib.MEM_DEVICE))
        return buf.view((self.primitives, self.classes), np.float32)

    # ---- new functionality: raw accumulator access for the cross-GPU sum ----------
    def get_raw(self):
        out = np.empty((self.primitives, self.classes), np.float32)
        if out.size:
            _lib.check(_lib.lib().smesh_aggregator_get_raw(self._h, out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))
        return out

    def set_raw(self, raw):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        if raw.shape != (self.primitives, self.classes):
            raise ValueError("raw accumulator must be float32[%d,%d]" % (self.primitives, self.classes))
        if raw.size:
            _lib.check(_lib.lib().smesh_aggregator_set_raw(self._h, raw.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))

    def raw_device_array(self, padded=False):
        """The un-normalised accumulator in HBM (a view, not a copy).  Rows are padded to `row_stride`
        floats in device memory: by default a strided (P,C) view is returned; `padded=True` gives the flat
        float32[P*row_stride] buffer (padding is zero), which is what an in-place all-reduce sums."""
        p, n, s = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint32()
        _lib.check(_lib.lib().smesh_aggregator_raw_pointer(self._h, ctypes.byref(p), ctypes.byref(n)))
        _lib.check(_lib.lib().smesh_aggregator_row_stride(self._h, ctypes.byref(s)))
        if padded:
            return DeviceArray(p.value, (int(n.value),), np.float32, self.device, owner=self)
        return DeviceArray(p.value, (self.primitives, self.classes), np.float32, self.device,
                           strides=(int(s.value), 1), owner=self)

    def renderer(self):
        """`ModelAggregator::renderer()` (Mesh.h:124-129): snapshot of the fused annotations for image gathers."""
        return ModelRenderer(self)

    def fuse_view(self, renderer, camera, probs_image, weights_image=None):
        """render(camera) + add(indices, probs) in one call without the indices leaving the device."""
        W, H = camera.resolution
        if (W > 0 and H > 0 and W <= 65536 and H <= 65536 and W * H < 0x7FFFFFFF // 4 and renderer.device == self.device
                and self._deferrable(probs_image, weights_image, W, H)):
            self._defer(renderer, _lib.CameraPOD.from_buffer_copy(camera._pod), W, H, probs_image, weights_image)
            return
        streams = []
        pp, pmem, pshape, pdt, pstr, k1 = describe(probs_image, 3, "probs image", self.device, streams)
        if tuple(pshape) != (W, H, self.classes) or pdt != np.float32:
            raise ValueError("probs image must be float32 (W,H,C) = %s" % ((W, H, self.classes),))
        if pstr != (H * self.classes, self.classes, 1):
            raise ValueError("fuse_view needs a contiguous (W,H,C) probs image")
        wp = None
        if weights_image is not None:
            wp_, wmem, wshape, wdt, wstr, k2 = describe(weights_image, 2, "weights image", self.device, streams)
            if tuple(wshape) != (W, H) or wdt != np.float32 or wstr != (H, 1) or wmem != pmem:
                raise ValueError("weights image must be contiguous float32 (W,H) in the same memory as probs")
            wp = ctypes.c_void_p(wp_)
        _lib.check(_lib.lib().smesh_fuse_view(renderer._h, self._h, ctypes.byref(camera._pod), ctypes.c_void_p(pp), wp, pmem))
        release_to(self.device, streams)
        if pmem == _lib.MEM_DEVICE:
            self._hold([k1, k2 if weights_image is not None else None])

    def _marshal_views(self, cameras, probs_images, weights_images, what):
        """ctypes arguments of a batch of views: (pods, n, probs pointers, weights pointers or None, memory kind, keep-alives, streams)."""
        cameras, probs_images = list(cameras), list(probs_images)
        n = len(cameras)
        if len(probs_images) != n or (weights_images is not None and len(weights_images) != n):
            raise ValueError("%s needs one probs image (and one weights image or None) per camera" % what)
        pods = (_lib.CameraPOD * max(n, 1))()
        pptr, wptr = (ctypes.c_void_p * max(n, 1))(), (ctypes.c_void_p * max(n, 1))()
        keep, mem, streams = [], None, []
        for i, cam in enumerate(cameras):
            W, H = cam.resolution
            pods[i] = cam._pod
            pp, pmem, pshape, pdt, pstr, k1 = describe(probs_images[i], 3, "probs image", self.device, streams)
            if tuple(pshape) != (W, H, self.classes) or pdt != np.float32:
                raise ValueError("probs image %d must be float32 (W,H,C) = %s" % (i, (W, H, self.classes)))
            if pstr != (H * self.classes, self.classes, 1):
                raise ValueError("%s needs contiguous (W,H,C) probs images" % what)
            if mem is None:
                mem = pmem
            if pmem != mem:
                raise ValueError("%s: all images must live in the same memory (host or device)" % what)
            pptr[i] = pp
            keep.append(k1)
            w = None if weights_images is None else weights_images[i]
            if w is not None:
                wp_, wmem, wshape, wdt, wstr, k2 = describe(w, 2, "weights image", self.device, streams)
                if tuple(wshape) != (W, H) or wdt != np.float32 or wstr != (H, 1) or wmem != mem:
                    raise ValueError("weights image %d must be contiguous float32 (W,H) in the same memory as probs" % i)
                wptr[i] = wp_
                keep.append(k2)
        return pods, n, pptr, (None if weights_images is None else wptr), (mem if mem is not None else _lib.MEM_HOST), keep, streams

    def fuse_views(self, renderer, cameras, probs_images, weights_images=None):
        """`fuse_view` for a whole batch, in order (the loop of colorize_cityscapes_mesh.py:54-67 as one call).  With a
        triangle renderer and device-resident images the library rasterises and fuses up to eight views per launch: each
        accumulator row is read and written once for all of them.  All images must live in the same memory (host or device)."""
        pods, n, pptr, wptr, mem, keep, streams = self._marshal_views(cameras, probs_images, weights_images, "fuse_views")
        if n == 0:
            return
        _lib.check(_lib.lib().smesh_fuse_views(renderer._h, self._h, pods, n, pptr, wptr, mem))
        release_to(self.device, streams)
        if mem == _lib.MEM_DEVICE:
            self._hold(keep)

    def fuse_views_ranged(self, renderer, cameras, probs_images, weights_images=None, nparts=4, on_rows=None):
        """`fuse_views` cut by accumulator row range (new functionality; `smesh_fuse_views_begin` / `_continue`): all
        views (at most 32) are rasterised, then part p = 0 .. nparts-1 fuses, for all of them in order, the triangles whose rows lie
        in one 64-row-aligned range, and `on_rows(row_lo, row_hi)` is called as soon as that part is queued -- those rows are final,
        so a sharded job exchanges them (`Communicator.allreduce_rows`) while the next part is fused.  Same sums as `fuse_views`.
        Where rows are not in triangle order (texel renderers, re-ordered meshes, host images ...) part 0 is the whole job.
        Returns the list of (row_lo, row_hi)."""
        pods, n, pptr, wptr, mem, keep, streams = self._marshal_views(cameras, probs_images, weights_images, "fuse_views_ranged")
        nparts = int(nparts)
        if n == 0 or nparts < 1:
            raise ValueError("fuse_views_ranged needs at least one view and nparts >= 1")
        lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
        ranges = []
        lib = _lib.lib()
        _lib.check(lib.smesh_fuse_views_begin(renderer._h, self._h, pods, n, pptr, wptr, mem, nparts, ctypes.byref(lo), ctypes.byref(hi)))
        for part in range(nparts):
            if part:
                _lib.check(lib.smesh_fuse_views_continue(renderer._h, self._h, part, ctypes.byref(lo), ctypes.byref(hi)))
            ranges.append((int(lo.value), int(hi.value)))
            if on_rows is not None and hi.value > lo.value:
                on_rows(int(lo.value), int(hi.value))
        release_to(self.device, streams)
        if mem == _lib.MEM_DEVICE:
            self._hold(keep)
        return ranges

    def get_raw_rows(self, row_lo, row_hi, plane=0):
        """Rows [row_lo, row_hi) of one plane of the raw state (plane 0: the accumulator as stored -- Mul: its hi plane, unfolded;
        plane 1: Mul's lo plane; a Mul element's value is hi + lo): float32[row_hi - row_lo, C]."""
        row_lo, row_hi = int(row_lo), int(row_hi)
        out = np.empty((max(row_hi - row_lo, 0), self.classes), np.float32)
        _lib.check(_lib.lib().smesh_aggregator_get_raw_rows(self._h, row_lo, row_hi, int(plane), out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))
        return out

    def set_raw_rows(self, row_lo, raw, plane=0):
        raw = np.ascontiguousarray(raw, dtype=np.float32)
        if raw.ndim != 2 or raw.shape[1] != self.classes:
            raise ValueError("raw rows must be float32[n,%d]" % self.classes)
        _lib.check(_lib.lib().smesh_aggregator_set_raw_rows(self._h, int(row_lo), int(row_lo) + raw.shape[0], int(plane),
                                                           raw.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))


class ModelRenderer:
    """Fused annotations gathered back to an image: `ModelAggregator::renderer()` + `ModelRenderer::render`
    (/root/reference/include/semantic_meshes/fusion/Mesh.h:124-129, 25-42).  Holds a snapshot of get()."""

    def __init__(self, aggregator):
        self.classes, self.device = aggregator.classes, aggregator.device
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_aggregator_renderer(aggregator._h, ctypes.byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                _lib.lib().smesh_annotation_renderer_destroy(h)
            except Exception:
                pass

    def render(self, primitive_image, background=None):
        """float32 (W,H,C) image: annotation of the primitive under each pixel, `background` (C floats,
        default zeros) where the index is out of range."""
        ip, imem, ishape, idt, istr, keep = describe(primitive_image, 2, "primitive image", self.device)
        if idt not in _IDX_CODES:
            raise ValueError("primitive image dtype must be one of uint32/int32/uint64/int64, got %s" % idt)
        bg = np.zeros(self.classes, np.float32) if background is None else np.ascontiguousarray(background, dtype=np.float32)
        if bg.shape != (self.classes,):
            raise ValueError("background must have %d entries" % self.classes)
        W, H = ishape
        out = np.empty((W, H, self.classes), np.float32)
        if out.size:
            _lib.check(_lib.lib().smesh_annotation_renderer_render(
                self._h, ctypes.c_void_p(ip), _IDX_CODES[idt], _c64(istr), imem, bg.ctypes.data_as(ctypes.c_void_p),
                out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST, W, H))
        return out


    def render_device(self, primitive_image, background=None):
        """`render()` with the (W,H,C) image left in HBM: a `DeviceArray` (`__cuda_array_interface__` / DLPack) in a fresh allocation
        owned by the returned object -- what the harness's `tf.gather(annotations, idx)` (eval_scannet.py:314) produces."""
        from .device import DeviceBuffer
        ip, imem, ishape, idt, istr, keep = describe(primitive_image, 2, "primitive image", self.device)
        if idt not in _IDX_CODES:
            raise ValueError("primitive image dtype must be one of uint32/int32/uint64/int64, got %s" % idt)
        bg = np.zeros(self.classes, np.float32) if background is None else np.ascontiguousarray(background, dtype=np.float32)
        if bg.shape != (self.classes,):
            raise ValueError("background must have %d entries" % self.classes)
        W, H = ishape
        buf = DeviceBuffer(max(W * H * self.classes * 4, 4), self.device)
        if W * H:
            _lib.check(_lib.lib().smesh_annotation_renderer_render(
                self._h, ctypes.c_void_p(ip), _IDX_CODES[idt], _c64(istr), imem, bg.ctypes.data_as(ctypes.c_void_p),
                ctypes.c_void_p(buf.ptr), _lib.MEM_DEVICE, W, H))
        return buf.view((W, H, self.classes), np.float32)


class MeshAggregatorSum(_MeshAggregator):
    pass


class MeshAggregatorSummax(_MeshAggregator):
    pass


class MeshAggregatorMul(_MeshAggregator):
    pass


_CLASSES = {"Sum": MeshAggregatorSum, "Summax": MeshAggregatorSummax, "Mul": MeshAggregatorMul}


def MeshAggregator(primitives, classes, aggregator="sum", images_equal_weight=0.5, device=0):
    """`semantic_meshes.fusion.MeshAggregator(primitives, classes[, aggregator[, images_equal_weight]])`
    (Fusion.cu:120-138,148-150): aggregator name is matched after capitalising its first letter."""
    name = str(aggregator)
    name = name[:1].upper() + name[1:]
    if name not in _CLASSES:
        raise ValueError("unknown aggregator %r (expected one of sum, summax, mul)" % (aggregator,))
    return _CLASSES[name](primitives, classes, name, images_equal_weight, device)
