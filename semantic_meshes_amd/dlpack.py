"""Minimal DLPack (v0.x `dltensor` capsule) producer/consumer in ctypes.

The reference hands render() results to Python as DLPack capsules over device memory and accepts
capsules/tensors in add() (/root/reference/python/semantic_meshes/include/Renderer.h:37-38, Common.h:5-30).
`DeviceArray.__dlpack__()` exports HBM buffers (kDLROCM) so that e.g. `torch.from_dlpack(indices)` is
zero-copy; `from_capsule()` lets add() consume objects that only speak DLPack.
"""
import ctypes

import numpy as np

kDLCPU, kDLCUDA, kDLCUDAHost, kDLROCM, kDLROCMHost = 1, 2, 3, 10, 11
_CODES = {"i": 0, "u": 1, "f": 2}
_KINDS = {0: "i", 1: "u", 2: "f"}


class DLDevice(ctypes.Structure):
    _fields_ = [("device_type", ctypes.c_int32), ("device_id", ctypes.c_int32)]


class DLDataType(ctypes.Structure):
    _fields_ = [("code", ctypes.c_uint8), ("bits", ctypes.c_uint8), ("lanes", ctypes.c_uint16)]


class DLTensor(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("device", DLDevice), ("ndim", ctypes.c_int32), ("dtype", DLDataType),
                ("shape", ctypes.POINTER(ctypes.c_int64)), ("strides", ctypes.POINTER(ctypes.c_int64)),
                ("byte_offset", ctypes.c_uint64)]


class DLManagedTensor(ctypes.Structure):
    pass


_DELETER = ctypes.CFUNCTYPE(None, ctypes.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", ctypes.c_void_p), ("deleter", _DELETER)]

_api = ctypes.pythonapi
_api.PyCapsule_New.restype = ctypes.py_object
_api.PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
_api.PyCapsule_GetPointer.restype = ctypes.c_void_p
_api.PyCapsule_GetPointer.argtypes = [ctypes.py_object, ctypes.c_char_p]
_api.PyCapsule_IsValid.restype = ctypes.c_int
_api.PyCapsule_IsValid.argtypes = [ctypes.py_object, ctypes.c_char_p]
_api.PyCapsule_SetName.restype = ctypes.c_int
_api.PyCapsule_SetName.argtypes = [ctypes.py_object, ctypes.c_char_p]

_live = {}   # address of the DLManagedTensor -> everything that must outlive the consumer


@_DELETER
def _deleter(ptr):
    _live.pop(ctypes.addressof(ptr.contents), None)


_CAPSULE_DTOR = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


@_CAPSULE_DTOR
def _capsule_destructor(capsule):
    # an unconsumed capsule still carries the name "dltensor": release what it holds
    cap = ctypes.cast(capsule, ctypes.py_object)
    if _api.PyCapsule_IsValid(cap, b"dltensor"):
        addr = _api.PyCapsule_GetPointer(cap, b"dltensor")
        _live.pop(addr, None)


def to_capsule(ptr, shape, strides_elems, dtype, device_type, device_id, owner):
    """PyCapsule named "dltensor" over a DLManagedTensor describing the buffer; `owner` is kept alive
    until the consumer calls the deleter."""
    dtype = np.dtype(dtype)
    nd = len(shape)
    shp = (ctypes.c_int64 * nd)(*shape)
    std = (ctypes.c_int64 * nd)(*strides_elems)
    mt = DLManagedTensor()
    mt.dl_tensor.data = ctypes.c_void_p(ptr)
    mt.dl_tensor.device = DLDevice(device_type, device_id)
    mt.dl_tensor.ndim = nd
    mt.dl_tensor.dtype = DLDataType(_CODES[dtype.kind], dtype.itemsize * 8, 1)
    mt.dl_tensor.shape = shp
    mt.dl_tensor.strides = std
    mt.dl_tensor.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _deleter
    addr = ctypes.addressof(mt)
    _live[addr] = (mt, shp, std, owner)
    return _api.PyCapsule_New(addr, b"dltensor", ctypes.cast(_capsule_destructor, ctypes.c_void_p))


class Imported:
    """A consumed capsule: pointer, shape, dtype, element strides, memory kind; releases on `close()`."""

    def __init__(self, capsule):
        self._mt = None
        if not _api.PyCapsule_IsValid(capsule, b"dltensor"):
            raise ValueError("expected an unconsumed DLPack capsule named 'dltensor'")
        addr = _api.PyCapsule_GetPointer(capsule, b"dltensor")
        self._mt = ctypes.cast(addr, ctypes.POINTER(DLManagedTensor))
        _api.PyCapsule_SetName(capsule, b"used_dltensor")
        self._capsule = capsule
        t = self._mt.contents.dl_tensor
        if t.dtype.lanes != 1 or t.dtype.code not in _KINDS:
            self.close()
            raise ValueError("unsupported DLPack dtype")
        self.dtype = np.dtype("%s%d" % (_KINDS[t.dtype.code], t.dtype.bits // 8))
        self.shape = tuple(int(t.shape[i]) for i in range(t.ndim))
        if t.strides:
            self.strides = tuple(int(t.strides[i]) for i in range(t.ndim))
        else:
            st, acc = [], 1
            for s in reversed(self.shape):
                st.append(acc)
                acc *= s
            self.strides = tuple(reversed(st))
        self.ptr = (t.data or 0) + int(t.byte_offset)
        self.device_type, self.device_id = int(t.device.device_type), int(t.device.device_id)
        self.on_device = self.device_type in (kDLCUDA, kDLROCM)
        if self.device_type not in (kDLCPU, kDLCUDA, kDLCUDAHost, kDLROCM, kDLROCMHost):
            self.close()
            raise ValueError("unsupported DLPack device type %d" % self.device_type)

    def close(self):
        mt, self._mt = getattr(self, "_mt", None), None
        if mt is not None and mt.contents.deleter:
            mt.contents.deleter(mt)

    def __del__(self):
        self.close()


def own_capsule_owner(capsule):
    """If `capsule` is an UNCONSUMED capsule made by to_capsule() in this process, consume it and return the object it was made
    over (nobody else has seen the memory); else None."""
    if type(capsule).__name__ != "PyCapsule" or not _api.PyCapsule_IsValid(capsule, b"dltensor"):
        return None
    addr = _api.PyCapsule_GetPointer(capsule, b"dltensor")
    entry = _live.get(addr)
    if entry is None:
        return None
    owner = entry[3]
    _api.PyCapsule_SetName(capsule, b"used_dltensor")
    _live.pop(addr, None)
    return owner
