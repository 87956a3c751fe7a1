"""ctypes loader for the HIP library (semantic_meshes_amd/csrc/libsmesh_hip.so).

The C ABI is declared in include/smesh.h.  There is deliberately NO CPU fallback here: if the
HIP library is missing or no GPU is usable, calls raise -- the product never routes through oracle/.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMESH_LIB_PATH") or os.path.join(_HERE, "csrc", "libsmesh_hip.so")   # (SMESH_LIB_PATH: a development build, e.g. `make ABLATION=1` in a copy of csrc/)

OK, ERR_INVALID, ERR_RUNTIME, ERR_NODEVICE = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
IDX_U32, IDX_I32, IDX_U64, IDX_I64 = 0, 1, 2, 3
AGG_KINDS = {"Sum": 0, "Summax": 1, "Mul": 2}
PROF_FUSE_SCATTER, PROF_FUSE_HIST, PROF_RASTER, PROF_FINALIZE, PROF_EXCHANGE = 0, 1, 2, 3, 4

c_void_p, c_int, c_u64, c_u32, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_float
P = ctypes.POINTER


class CameraPOD(ctypes.Structure):
    """smesh_camera_t (include/smesh.h)."""
    _fields_ = [
        ("rotation", ctypes.c_float * 9),
        ("translation", ctypes.c_float * 3),
        ("focal", ctypes.c_double * 2),
        ("principal", ctypes.c_double * 2),
        ("width", ctypes.c_uint64),
        ("height", ctypes.c_uint64),
    ]


# name -> (restype, argtypes); one entry per symbol declared in include/smesh.h
SIGNATURES = {
    "smesh_backend": (ctypes.c_char_p, []),
    "smesh_last_error": (ctypes.c_char_p, []),
    "smesh_device_count": (c_int, [P(c_int)]),
    "smesh_synchronize": (c_int, [c_int]),
    "smesh_stream_wait": (c_int, [c_int, c_void_p]),
    "smesh_stream_release": (c_int, [c_int, c_void_p]),
    "smesh_stream_handle": (c_int, [c_int, P(c_void_p)]),
    "smesh_token_record": (c_int, [c_int, P(c_u64)]),
    "smesh_token_done": (c_int, [c_int, c_u64, P(c_int)]),
    "smesh_stream_mark": (c_int, [c_int, c_int]),
    "smesh_stream_mark_elapsed": (c_int, [c_int, c_int, c_int, P(ctypes.c_double)]),
    "smesh_renderer_create_triangles": (c_int, [c_void_p, c_u64, c_void_p, c_u64, c_int, P(c_void_p)]),
    "smesh_renderer_create_texels": (c_int, [c_void_p, c_u64, c_void_p, c_u64, c_void_p, c_u64, c_float, c_int, P(c_void_p)]),
    "smesh_renderer_destroy": (c_int, [c_void_p]),
    "smesh_renderer_num_primitives": (c_int, [c_void_p, P(c_u64)]),
    "smesh_renderer_texel_layout": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "smesh_renderer_render": (c_int, [c_void_p, P(CameraPOD), c_void_p, c_void_p]),
    "smesh_renderer_render_device": (c_int, [c_void_p, P(CameraPOD), P(c_void_p), P(c_void_p)]),
    "smesh_renderer_release_image": (c_int, [c_void_p, c_void_p, c_void_p]),
    "smesh_aggregator_create": (c_int, [c_u64, c_u32, c_int, c_float, c_int, P(c_void_p)]),
    "smesh_aggregator_destroy": (c_int, [c_void_p]),
    "smesh_aggregator_reset": (c_int, [c_void_p]),
    "smesh_aggregator_add": (c_int, [c_void_p, c_void_p, c_int, P(ctypes.c_int64), c_int,
                                     c_void_p, P(ctypes.c_int64), c_int,
                                     c_void_p, P(ctypes.c_int64), c_int, c_u64, c_u64]),
    "smesh_aggregator_add_async": (c_int, [c_void_p, c_void_p, c_int, P(ctypes.c_int64), c_int,
                                           c_void_p, P(ctypes.c_int64), c_int,
                                           c_void_p, P(ctypes.c_int64), c_int, c_u64, c_u64]),
    "smesh_aggregator_add_many": (c_int, [c_void_p, c_u64, P(c_void_p), c_int, P(ctypes.c_int64), c_int,
                                          P(c_void_p), P(ctypes.c_int64), c_int,
                                          P(c_void_p), P(ctypes.c_int64), c_int, c_u64, c_u64]),
    "smesh_aggregator_add_rendered": (c_int, [c_void_p, c_void_p, c_void_p,
                                              c_void_p, P(ctypes.c_int64), c_int,
                                              c_void_p, P(ctypes.c_int64), c_int, c_u64, c_u64]),
    "smesh_aggregator_get": (c_int, [c_void_p, c_void_p, c_int]),
    "smesh_aggregator_get_rows": (c_int, [c_void_p, c_u64, c_u64, c_void_p, c_int]),
    "smesh_aggregator_get_raw": (c_int, [c_void_p, c_void_p, c_int]),
    "smesh_aggregator_set_raw": (c_int, [c_void_p, c_void_p, c_int]),
    "smesh_aggregator_raw_pointer": (c_int, [c_void_p, P(c_void_p), P(c_u64)]),
    "smesh_aggregator_row_stride": (c_int, [c_void_p, P(c_u32)]),
    "smesh_comm_unique_id": (c_int, [c_void_p]),
    "smesh_comm_create": (c_int, [c_int, c_int, c_int, c_void_p, P(c_void_p)]),
    "smesh_comm_create_all": (c_int, [P(c_int), c_int, P(c_void_p)]),
    "smesh_comm_destroy": (c_int, [c_void_p]),
    "smesh_comm_rank": (c_int, [c_void_p, P(c_int), P(c_int)]),
    "smesh_allreduce": (c_int, [P(c_void_p), P(c_void_p), c_int]),
    "smesh_reduce_scatter": (c_int, [c_void_p, c_void_p, P(c_u64), P(c_u64)]),
    "smesh_fuse_views_begin": (c_int, [c_void_p, c_void_p, c_void_p, c_u64, P(c_void_p), P(c_void_p), c_int, c_int, P(c_u64), P(c_u64)]),
    "smesh_fuse_views_continue": (c_int, [c_void_p, c_void_p, c_int, P(c_u64), P(c_u64)]),
    "smesh_allreduce_rows": (c_int, [c_void_p, c_void_p, c_u64, c_u64]),
    "smesh_exchange_join": (c_int, [c_void_p]),
    "smesh_aggregator_get_raw_rows": (c_int, [c_void_p, c_u64, c_u64, c_int, c_void_p, c_int]),
    "smesh_aggregator_set_raw_rows": (c_int, [c_void_p, c_u64, c_u64, c_int, c_void_p, c_int]),
    "smesh_comm_allreduce_f64": (c_int, [c_void_p, P(ctypes.c_double), c_int, c_int]),
    "smesh_aggregator_renderer": (c_int, [c_void_p, P(c_void_p)]),
    "smesh_annotation_renderer_render": (c_int, [c_void_p, c_void_p, c_int, P(ctypes.c_int64), c_int, c_void_p, c_void_p, c_int, c_u64, c_u64]),
    "smesh_annotation_renderer_destroy": (c_int, [c_void_p]),
    "smesh_fuse_view": (c_int, [c_void_p, c_void_p, P(CameraPOD), c_void_p, c_void_p, c_int]),
    "smesh_fuse_views": (c_int, [c_void_p, c_void_p, P(CameraPOD), c_u64, P(c_void_p), P(c_void_p), c_int]),
    "smesh_aggregator_add_matched": (c_int, [c_void_p, c_void_p, c_void_p, c_int, P(ctypes.c_int64), c_int,
                                             c_void_p, P(ctypes.c_int64), c_int,
                                             c_void_p, P(ctypes.c_int64), c_int, c_u64, c_u64, P(c_int)]),
    "smesh_renderer_seal_render": (c_int, [c_void_p, c_void_p]),
    "smesh_renderer_render_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "smesh_box_extent_bound": (c_int, [c_void_p, c_u64, c_void_p, c_u64, c_void_p, c_void_p]),
    "smesh_last_fuse_kernel": (ctypes.c_char_p, []),
    "smesh_last_add_path": (ctypes.c_char_p, []),
    "smesh_profile_enable": (c_int, [c_int, c_int]),
    "smesh_profile_sample_every": (c_int, [c_int, ctypes.c_uint32]),
    "smesh_profile_read": (c_int, [c_int, c_int, P(ctypes.c_double), P(c_u64)]),
    "smesh_profile_read_ex": (c_int, [c_int, c_int, P(ctypes.c_double), P(c_u64), P(c_u64), P(c_u64)]),
    "smesh_profile_regions": (c_int, [c_int, c_int, P(c_u64)]),
    "smesh_profile_reset": (c_int, [c_int]),
    "smesh_synth_probs": (c_int, [c_void_p, c_u64, c_u32, c_u64, c_float, c_int, c_int]),
    "smesh_device_malloc": (c_int, [c_int, c_u64, P(c_void_p)]),
    "smesh_device_free": (c_int, [c_int, c_void_p]),
    "smesh_device_trim": (c_int, [c_int, P(c_u64)]),
    "smesh_set_option": (c_int, [ctypes.c_char_p, ctypes.c_int64]),
    "smesh_get_option": (c_int, [ctypes.c_char_p, P(ctypes.c_int64)]),
    "smesh_host_malloc": (c_int, [c_u64, P(c_void_p)]),
    "smesh_host_free": (c_int, [c_void_p]),
    "smesh_memcpy": (c_int, [c_void_p, c_void_p, c_u64, c_int, c_int, c_int]),
}

_lib = None
_lock = threading.Lock()


def _elf_dynamic_strings(path):
    """(SONAME or None, [DT_NEEDED ...]) of a little-endian ELF64 shared object, read from its PT_DYNAMIC segment through the program
    headers (section headers may be stripped).  Raises ValueError on anything that is not such a file, truncated ones included."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    try:
        if data[:6] != b"\x7fELF\x02\x01":
            raise ValueError("not a little-endian ELF64 file")
        phoff, = struct.unpack_from("<Q", data, 0x20)
        phentsize, phnum = struct.unpack_from("<HH", data, 0x36)
        loads, dyn = [], None
        for k in range(phnum):
            typ, flags, off, vaddr, paddr, filesz, memsz, align = struct.unpack_from("<IIQQQQQQ", data, phoff + k * phentsize)
            if typ == 1:
                loads.append((vaddr, off, filesz))
            elif typ == 2:
                dyn = (off, filesz)
        if dyn is None:
            return None, []
        entries = []
        for e in range(dyn[0], dyn[0] + dyn[1], 16):
            tag, val = struct.unpack_from("<qQ", data, e)
            if tag == 0:
                break
            entries.append((tag, val))
        strtab = next((v for t, v in entries if t == 5), None)        # DT_STRTAB: a virtual address
        if strtab is None:
            return None, []
        stroff = next((off + strtab - vaddr for vaddr, off, filesz in loads if vaddr <= strtab < vaddr + filesz), None)
        if stroff is None:
            raise ValueError("DT_STRTAB outside every PT_LOAD segment")

        def cstr(o):
            end = data.index(b"\0", stroff + o)
            return data[stroff + o:end].decode()

        soname, needed = None, []
        for tag, val in entries:
            if tag == 1:
                needed.append(cstr(val))
            elif tag == 14:
                soname = cstr(val)
        return soname, needed
    except (struct.error, IndexError, UnicodeDecodeError) as e:
        raise ValueError("malformed ELF file %s: %s" % (path, e))


def _preload_note(msg):
    """Why the torch-bundled HIP runtime was NOT mapped first (SMESH_DEBUG=1 prints it; a process that later imports torch against
    a second runtime sees no GPU there -- this is the first thing to look at)."""
    global PRELOAD_NOTE
    PRELOAD_NOTE = msg
    if os.environ.get("SMESH_DEBUG"):
        import sys
        print("semantic_meshes_amd: " + msg, file=sys.stderr)


PRELOAD_NOTE = None


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels ship their own libamdhip64 / libhsa-runtime64 next to torch; a
    process that first loads this library against the system ROCm and later imports torch ends up with TWO runtimes, and
    the second one sees no usable GPU (`torch.cuda.is_available()` False, a hang at exit).  The other order is fine: the
    dynamic linker satisfies our DT_NEEDED `libamdhip64.so.N` with the copy torch already mapped -- IF that copy's SONAME
    is the same `libamdhip64.so.N`.  So, when a torch installation exists in this environment (located, NOT imported) AND
    its bundled runtime carries exactly the SONAME this library was linked against, that runtime is mapped first and both
    then share it in any import order.  A wheel built against another ROCm major (its SONAME differs) is left alone: the
    linker could not reuse it, and mapping it would CREATE the two-runtime state for processes that never import torch.
    SMESH_HIP_RUNTIME=system keeps the system ROCm unconditionally; =<directory> takes libamdhip64.so / libhsa-runtime64.so
    from there (same SONAME check)."""
    import importlib.util
    import sys
    choice = os.environ.get("SMESH_HIP_RUNTIME", "auto")
    if choice == "system" or "torch" in sys.modules:
        return
    libdir = choice if choice not in ("auto", "torch") else None
    if libdir is None:
        try:
            spec = importlib.util.find_spec("torch")
        except (ImportError, ValueError):
            spec = None
        if spec is None or not spec.origin:
            return
        libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    hip = os.path.join(libdir, "libamdhip64.so")
    if not os.path.exists(hip):
        return
    try:
        wanted = [n for n in _elf_dynamic_strings(LIB_PATH)[1] if n.startswith("libamdhip64.so")]
        offered = _elf_dynamic_strings(hip)[0]
    except (OSError, ValueError) as e:
        _preload_note("HIP runtime preload skipped: %s" % e)
        return
    if not wanted or offered != wanted[0]:
        _preload_note("HIP runtime preload skipped: %s offers SONAME %r, libsmesh_hip.so needs %r" % (hip, offered, wanted[:1]))
        return          # another ROCm major: the linker would map the system runtime beside it anyway
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return
    # ... and the RCCL that goes with that runtime (comm.cpp loads RCCL lazily; torch would otherwise map a second one later)
    rccl = os.path.join(libdir, "librccl.so")
    if os.path.exists(rccl):
        os.environ.setdefault("SMESH_RCCL_LIB", rccl)


def lib():
    """Load libsmesh_hip.so (once).  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        "semantic_meshes_amd: %s not found -- build it with `make -C semantic_meshes_amd/csrc` "
                        "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
                _preload_hip_runtime()
                L = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)  # AttributeError if the library does not export the ABI
                    fn.restype = res
                    fn.argtypes = args
                _lib = L
    return _lib


def check(status):
    """Map a C status to the reference's exception types (ValueError <- std::invalid_argument)."""
    if status == OK:
        return
    msg = lib().smesh_last_error().decode(errors="replace")
    if status == ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def device_count():
    n = c_int(0)
    check(lib().smesh_device_count(ctypes.byref(n)))
    return n.value


_flush_hooks = []      # callables(device): work the Python layer has accepted but not yet handed to the library (fusion.py: deferred views)


def flush_pending(device=None):
    """Hand every deferred view (MeshAggregator.add / fuse_view, fusion.py) of `device` (None: all devices) to the library now."""
    for hook in list(_flush_hooks):
        hook(device)


def last_fuse_kernel():
    """Name of the fusion kernel the calling thread's last add / fuse call used (`smesh_last_fuse_kernel`) -- deferred views
    (fusion.py) are handed to the library first, so that "last call" means the caller's last call."""
    flush_pending()
    return lib().smesh_last_fuse_kernel().decode()


def last_add_path():
    """Which path the calling thread's last add() took (`smesh_last_add_path`); deferred views are handed over first."""
    flush_pending()
    return lib().smesh_last_add_path().decode()


def synchronize(device=0):
    flush_pending(device)
    check(lib().smesh_synchronize(device))
