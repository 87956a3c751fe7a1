"""Device-resident arrays handed between render() and add().

The reference returns DLPack capsules over CUDA memory from render() and accepts them (or numpy
arrays, or framework tensors) in add() (/root/reference/python/semantic_meshes/include/Renderer.h:37-38,
Common.h:5-30).  Here the hand-off object is a DeviceArray: a typed, strided view of HIP device
memory that exposes `__cuda_array_interface__` (the protocol ROCm builds of torch/cupy consume)
and converts to numpy on request.
"""
import ctypes
import os

import numpy as np

from . import _lib


class DeviceArray:
    """A (possibly strided) view of device memory owned by some handle of this library."""

    def __init__(self, ptr, shape, dtype, device=0, strides=None, owner=None, on_release=None):
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        if strides is None:  # C-contiguous, in elements
            strides, acc = [], 1
            for s in reversed(self.shape):
                strides.append(acc)
                acc *= s
            strides = tuple(reversed(strides))
        self.strides = tuple(int(s) for s in strides)  # in ELEMENTS
        self._owner = owner            # keeps the owning handle alive
        self._on_release = on_release  # returns the buffer to its pool
        self._rendered_by = None       # render(): the renderer whose latest output this is (MeshAggregator.add fast path)
        self._exported = False         # handed to another framework: the contents may have been changed behind our back

    unrun = False                      # (render.py: a plane of a render() that has not been rasterised yet says True)

    def __del__(self):
        cb, self._on_release = getattr(self, "_on_release", None), None
        if cb is not None:
            try:
                cb(self.ptr)
            except Exception:
                pass

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def __cuda_array_interface__(self):
        # the consumer reads on ITS stream: everything this library still has in flight (render() is asynchronous on a
        # non-blocking stream) must have landed first -- same contract as __dlpack__ below
        self._seal()
        self._exported = True
        _lib.synchronize(self.device)
        return self._cai_dict()

    def _seal(self):
        """A render() result whose content is about to leave the library (export, host copy): let the renderer take its checksum
        now, while it is what the rasteriser wrote -- copies of it are then recognised by MeshAggregator.add()."""
        rb = self._rendered_by
        if rb is not None and not self._exported and getattr(rb, "_h", None) is not None and rb._h.value:
            _lib.check(_lib.lib().smesh_renderer_seal_render(rb._h, ctypes.c_void_p(self.ptr)))

    def _cai_dict(self):
        return {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "data": (self.ptr, False),
            "version": 2,
            "strides": tuple(s * self.dtype.itemsize for s in self.strides),
        }

    def __dlpack_device__(self):
        from . import dlpack
        return (dlpack.kDLROCM, self.device)

    def __dlpack__(self, stream=None, **kwargs):
        """DLPack capsule over the HBM buffer (what the reference's render() returns, Renderer.h:37-38).
        The producing stream is synchronised first: the consumer may use any stream."""
        from . import dlpack
        self._seal()
        self._exported = True
        _lib.synchronize(self.device)
        return dlpack.to_capsule(self.ptr, self.shape, self.strides, self.dtype, dlpack.kDLROCM, self.device, self)

    def capsule(self):
        """A `"dltensor"` PyCapsule over the buffer -- the object the reference's render() returns
        (python/semantic_meshes/include/Renderer.h:37-38) and `tf.experimental.dlpack.from_dlpack` / `torch.utils.dlpack.from_dlpack`
        take.  The capsule keeps this array alive; handed back unconsumed to MeshAggregator.add it is recognised as this array."""
        from . import dlpack
        self._seal()                       # whoever consumes the capsule may write to the plane later
        _lib.synchronize(self.device)      # ... and reads it on its own stream
        return dlpack.to_capsule(self.ptr, self.shape, self.strides, self.dtype, dlpack.kDLROCM, self.device, self)

    def transpose(self, *axes):
        """Stride permutation without a copy (callers transpose (H,W,C) network output to (W,H,C))."""
        if len(axes) == 1 and hasattr(axes[0], "__len__"):
            axes = tuple(axes[0])
        if not axes:
            axes = tuple(reversed(range(self.ndim)))
        return DeviceArray(self.ptr, [self.shape[a] for a in axes], self.dtype, self.device,
                           [self.strides[a] for a in axes], owner=self)

    @property
    def T(self):
        return self.transpose()

    def _is_contiguous(self):
        acc = 1
        for s, st in zip(reversed(self.shape), reversed(self.strides)):
            if s != 1 and st != acc:
                return False
            acc *= s
        return True

    def numpy(self):
        """Copy to a fresh host array (synchronises with the producing stream)."""
        self._seal()
        if not self._is_contiguous():
            # copy the dense span, then re-stride on the host
            span = 1 + sum((s - 1) * st for s, st in zip(self.shape, self.strides))
            flat = np.empty(span, self.dtype)
            _lib.check(_lib.lib().smesh_memcpy(flat.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr),
                                                flat.nbytes, _lib.MEM_HOST, _lib.MEM_DEVICE, self.device))
            return np.lib.stride_tricks.as_strided(flat, self.shape, [st * self.dtype.itemsize for st in self.strides]).copy()
        out = np.empty(self.shape, self.dtype)
        if out.nbytes:
            _lib.check(_lib.lib().smesh_memcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(self.ptr),
                                                out.nbytes, _lib.MEM_HOST, _lib.MEM_DEVICE, self.device))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%d, ptr=0x%x)" % (self.shape, self.dtype, self.device, self.ptr)


class DeviceBuffer:
    """Raw HBM allocation (benchmark inputs that must be resident before timing starts)."""

    def __init__(self, nbytes, device=0):
        self.nbytes, self.device = int(nbytes), int(device)
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_device_malloc(self.device, self.nbytes, ctypes.byref(p)))
        self.ptr = p.value

    def __del__(self):
        if getattr(self, "ptr", None):
            _lib.lib().smesh_device_free(self.device, ctypes.c_void_p(self.ptr))
            self.ptr = None

    def view(self, shape, dtype, offset_bytes=0):
        return DeviceArray(self.ptr + offset_bytes, shape, dtype, self.device, owner=self)


def trim(device=-1):
    """Unmap the device blocks the library keeps for reuse on `device` (-1: every device) and return how many bytes that was
    (`smesh_device_trim`).  Blocks the library's handles release stay mapped in a per-process cache (freshly mapped memory is where
    kernel writes were seen to go missing on a GPU shared by several processes; csrc/context.cpp) -- at most a sixteenth of the
    device's memory / 8 GiB per device (SMESH_ALLOC_CACHE_MB).  Call this before handing the GPU's memory to another allocator of
    the same process (torch, cupy) that needs the last gigabytes: live handles keep theirs, only idle blocks go."""
    held = ctypes.c_uint64(0)
    _lib.check(_lib.lib().smesh_device_trim(int(device), ctypes.byref(held)))
    return int(held.value)


def _free_pinned(ptr):
    try:
        _lib.lib().smesh_host_free(ctypes.c_void_p(ptr))
    except Exception:
        pass


def pinned_empty(shape, dtype=np.float32):
    """A numpy array over page-locked host memory (`smesh_host_malloc`): host images handed to add() from such an array
    cross PCIe by DMA at link speed.  The memory is released when the array and every view of it are gone."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    p = ctypes.c_void_p()
    _lib.check(_lib.lib().smesh_host_malloc(max(n, 1), ctypes.byref(p)))
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(p.value)   # numpy keeps `buf` alive as the base of every view
    weakref.finalize(buf, _free_pinned, p.value)
    return np.frombuffer(buf, dtype=dtype, count=n // dtype.itemsize).reshape(shape)


# Results (get(), get_rows(), get_raw()) land in page-locked host memory that is RECYCLED: a fresh pageable array of a cfg2 result
# (76 MB) costs 3.8 ms of first-touch page faults before a byte has crossed PCIe, a page-locked one is written by DMA at link speed
# (1.4 ms) -- but allocating page-locked memory is slow (tens of ms), so a buffer goes back to a free list when the array that was
# handed out (and every view of it) is gone, and the next result of that size takes it.  The caller still owns a fresh array each time.
_result_free = {}            # nbytes -> [pointers of free buffers]
_result_live = [0]           # page-locked bytes handed out or parked
_RESULT_MIN = 4 << 20
# (round 5: 32 GiB, so that cfg5's 12 GB result lands in page-locked memory too -- 438 ms through the staging ring into pageable
#  memory, 27 GB/s, against the DMA's ~55 GB/s; a host that cannot lock that much falls back to the ring, below)
_RESULT_LIMIT = int(os.environ.get("SMESH_RESULT_POOL_MB", "32768")) << 20
_RESULT_KEEP = 2             # free buffers kept per size (one for results of 1 GiB and more)


def _recycle_result(n, ptr):
    free = _result_free.setdefault(n, [])
    if len(free) < (1 if n >= (1 << 30) else _RESULT_KEEP):
        free.append(ptr)
    else:
        _result_live[0] -= n
        _free_pinned(ptr)


def result_empty(shape, dtype=np.float32):
    """An uninitialised numpy array for a result that is about to be copied out of HBM: page-locked and recycled when it is large
    (see above), plain `np.empty` otherwise or when SMESH_RESULT_POOL_MB worth of such arrays are alive."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if n < _RESULT_MIN or _RESULT_LIMIT <= 0:
        return np.empty(shape, dtype)
    free = _result_free.get(n)
    if free:
        ptr = free.pop()
    else:
        if _result_live[0] + n > _RESULT_LIMIT:
            return np.empty(shape, dtype)
        p = ctypes.c_void_p()
        try:
            _lib.check(_lib.lib().smesh_host_malloc(n, ctypes.byref(p)))
        except RuntimeError:
            return np.empty(shape, dtype)
        ptr = p.value
        _result_live[0] += n
    buf = (ctypes.c_uint8 * n).from_address(ptr)
    weakref.finalize(buf, _recycle_result, n, ptr)
    return np.frombuffer(buf, dtype=dtype, count=n // dtype.itemsize).reshape(shape)


def to_device(array, device=0):
    """Copy a host numpy array into a fresh device allocation; returns a DeviceArray owning it."""
    a = np.ascontiguousarray(array)
    buf = DeviceBuffer(max(a.nbytes, 1), device)
    if a.nbytes:
        _lib.check(_lib.lib().smesh_memcpy(ctypes.c_void_p(buf.ptr), a.ctypes.data_as(ctypes.c_void_p), a.nbytes,
                                            _lib.MEM_DEVICE, _lib.MEM_HOST, device))
    return buf.view(a.shape, a.dtype)


def _order_after_producer(obj, cai, device):
    """Device memory written by ANOTHER framework: order the library's (non-blocking) streams after the stream that
    produced it, without blocking the host.  The reference gets this for free from its synchronous cudaMemcpy on the null
    stream (Fusion.h:35-37).  Producer stream: torch tensors -> torch's current stream on that device; otherwise the
    `stream` entry of `__cuda_array_interface__` v3 (1 = legacy default, 2 = per-thread default, else a handle); absent
    -> the legacy default stream."""
    stream = 0
    if (type(obj).__module__ or "").split(".")[0] == "torch":
        try:
            import torch
            stream = int(torch.cuda.current_stream(obj.device).cuda_stream)
        except Exception:
            stream = 0
    elif cai is not None:
        s = cai.get("stream")
        stream = 0 if s in (None, 1) else int(s)   # 2 is also hipStreamPerThread's handle value
    _lib.check(_lib.lib().smesh_stream_wait(int(device or 0), ctypes.c_void_p(stream)))
    return stream


def release_to(device, streams):
    """After an asynchronous library call that read device arrays of other frameworks: their streams (`streams`, collected by
    describe()) wait for the library's reads, so the owners may overwrite or free the arrays right away -- no host wait."""
    for st in set(streams or ()):
        _lib.check(_lib.lib().smesh_stream_release(int(device), ctypes.c_void_p(st)))


def library_stream(device=0):
    """The library's main hipStream_t on `device` as an int (for `__dlpack__(stream=...)` / external stream wrappers)."""
    p = ctypes.c_void_p()
    _lib.check(_lib.lib().smesh_stream_handle(int(device), ctypes.byref(p)))
    return int(p.value or 0)


def describe(obj, want_ndim, what, device=None, streams=None):
    """Normalise an add()/render() argument to (pointer, memkind, shape, dtype, element strides, keepalive).

    Accepts what the reference's FromTensor accepts in practice: numpy arrays / array-likes
    on the host, and device arrays via DeviceArray or `__cuda_array_interface__` (torch-ROCm, cupy).
    Device memory that does not come from this library is ordered after its producer's stream (`device` = the GPU of
    the handle the argument is for); the producer streams are appended to `streams` for release_to().
    """
    if isinstance(obj, DeviceArray):
        shape, dtype, strides, ptr, mem, keep = obj.shape, obj.dtype, obj.strides, obj.ptr, _lib.MEM_DEVICE, obj
    elif hasattr(obj, "__cuda_array_interface__"):
        cai = obj.__cuda_array_interface__
        if device is not None:   # (None: layout bookkeeping only -- every entry point of the package passes its handle's GPU)
            st = _order_after_producer(obj, cai, device)
            if streams is not None:
                streams.append(st)
        shape, dtype = tuple(cai["shape"]), np.dtype(cai["typestr"])
        ptr = int(cai["data"][0])
        bstr = cai.get("strides")
        if bstr is None:
            strides, acc = [], 1
            for s in reversed(shape):
                strides.append(acc)
                acc *= s
            strides = tuple(reversed(strides))
        else:
            if any(b % dtype.itemsize for b in bstr):
                raise ValueError("%s: strides are not a multiple of the item size" % what)
            strides = tuple(b // dtype.itemsize for b in bstr)
        mem, keep = _lib.MEM_DEVICE, obj
    elif type(obj).__name__ == "PyCapsule" or (hasattr(obj, "__dlpack__") and not hasattr(obj, "__array_interface__")
                                                and not isinstance(obj, np.ndarray)):
        from . import dlpack
        if type(obj).__name__ == "PyCapsule":
            capsule = obj
            ordered = False
        else:
            try:   # DLPack protocol: the producer makes the data safe to read on the consumer's stream
                if device is None:
                    raise TypeError("no device to order on")
                capsule, ordered = obj.__dlpack__(stream=library_stream(device)), True
            except Exception:
                capsule, ordered = obj.__dlpack__(), False
        imp = dlpack.Imported(capsule)
        if imp.on_device and device is not None:
            if not ordered:
                st = _order_after_producer(obj, None, device)
            else:   # the producer ordered the hand-over itself; its later use of the buffer: assume its current / default stream
                # (it ordered the library's MAIN stream: the raster stream, where foreign index images are checksummed, follows)
                _lib.check(_lib.lib().smesh_stream_wait(int(device), ctypes.c_void_p(library_stream(device))))
                st = 0
                if (type(obj).__module__ or "").split(".")[0] == "torch":
                    try:
                        import torch
                        st = int(torch.cuda.current_stream(obj.device).cuda_stream)
                    except Exception:
                        st = 0
            if streams is not None:
                streams.append(st)
        shape, dtype, strides, ptr = imp.shape, imp.dtype, imp.strides, imp.ptr
        mem, keep = (_lib.MEM_DEVICE if imp.on_device else _lib.MEM_HOST), imp
    else:
        a = np.asarray(obj)
        if a.ndim == want_ndim and a.size:
            span = 1 + sum((s - 1) * abs(st) // a.itemsize for s, st in zip(a.shape, a.strides))
            bad = any(st < 0 or st % a.itemsize for st in a.strides) or span > 2 * a.size
            if bad or not a.flags.aligned:
                a = np.ascontiguousarray(a)
        shape, dtype = a.shape, a.dtype
        strides = tuple(st // a.itemsize for st in a.strides)
        ptr, mem, keep = a.ctypes.data, _lib.MEM_HOST, a
    if len(shape) != want_ndim:
        raise ValueError("%s must have rank %d, got shape %s" % (what, want_ndim, tuple(shape)))
    if mem == _lib.MEM_DEVICE and any(st < 0 for st in strides):
        raise ValueError("%s: negative strides are not supported for device arrays" % what)
    return ptr, mem, tuple(shape), dtype, strides, keep
