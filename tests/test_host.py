"""Host-side logic that needs no GPU: the Camera conversions, argument normalisation, PLY I/O, view sharding."""
import os

import numpy as np
import pytest

import semantic_meshes_amd as sm
from semantic_meshes_amd import _lib, device, distributed, synth


def test_camera_conversions_follow_the_reference_ctor():
    # python/semantic_meshes/include/Camera.h:16-57: rotation/translation -> float32; focal/principal -> float32 -> double
    R = np.eye(3, dtype=np.float64) * (1 + 1e-12)
    cam = sm.data.Camera(R, np.array([0.1, 0.2, 0.3]), np.array([640, 480], np.int64), np.array([512.123456789, 500.0]),
                         np.array([320.5, 240.25], np.float32))
    assert cam.rotation.dtype == np.float32 and cam.translation.dtype == np.float32
    assert cam.resolution == (640, 480) and cam.width == 640 and cam.height == 480
    assert cam.focal_lengths.dtype == np.float64 and cam.focal_lengths[0] == np.float64(np.float32(512.123456789))
    pod = cam._pod
    assert pod.width == 640 and pod.height == 480 and pod.focal[0] == float(np.float32(512.123456789))
    assert list(pod.rotation)[:3] == [1.0, 0.0, 0.0]


@pytest.mark.parametrize("bad", [
    dict(rotation=np.eye(4)), dict(translation=np.zeros(2)), dict(resolution=np.array([640.0, 480.0])),
    dict(resolution=np.array([0, 480])), dict(focal_lengths=np.zeros(3)), dict(rotation=np.eye(3, dtype=np.complex64)),
])
def test_camera_rejects_bad_arguments(bad):
    args = dict(rotation=np.eye(3), translation=np.zeros(3), resolution=np.array([640, 480]),
                focal_lengths=np.array([500.0, 500.0]), principal_point=np.array([320.0, 240.0]))
    args.update(bad)
    with pytest.raises(ValueError):
        sm.data.Camera(**args)


def test_describe_host_arrays():
    hwc = np.zeros((48, 64, 5), np.float32)
    ptr, mem, shape, dt, strides, keep = device.describe(hwc.transpose(1, 0, 2), 3, "probs")
    assert mem == _lib.MEM_HOST and shape == (64, 48, 5) and strides == (5, 64 * 5, 1) and ptr == hwc.ctypes.data
    # negative strides and sparse slices are copied
    ptr2, _, shape2, _, strides2, keep2 = device.describe(hwc[::-1].transpose(1, 0, 2), 3, "probs")
    assert strides2 == (48 * 5, 5, 1) and keep2.flags.c_contiguous
    _, _, _, _, strides3, keep3 = device.describe(hwc[::4].transpose(1, 0, 2), 3, "probs")
    assert keep3.flags.c_contiguous
    with pytest.raises(ValueError):
        device.describe(np.zeros((4, 4)), 3, "probs")


def test_describe_device_protocol():
    class Fake:
        __cuda_array_interface__ = {"shape": (8, 6), "typestr": "<u4", "data": (0x1000, False), "version": 2, "strides": (4, 32)}
    ptr, mem, shape, dt, strides, keep = device.describe(Fake(), 2, "idx")
    assert (ptr, mem, shape, dt, strides) == (0x1000, _lib.MEM_DEVICE, (8, 6), np.dtype(np.uint32), (1, 8))
    d = device.DeviceArray(0x2000, (8, 6, 3), np.float32)
    t = d.transpose(1, 0, 2)
    assert t.shape == (6, 8, 3) and t.strides == (3, 18, 1) and t.ptr == d.ptr and d.strides == (18, 3, 1)
    assert t._cai_dict()["strides"] == (12, 72, 4)   # (the property itself first waits for the library's stream: needs a GPU)


def test_aggregator_factory_names():
    # Fusion.cu:126: first letter capitalised, rest kept
    for bad in ("median", "SUM", "sumMax", ""):
        with pytest.raises((ValueError, RuntimeError)):
            sm.fusion.MeshAggregator(4, 3, aggregator=bad)


def test_ply_roundtrip(tmp_path):
    mesh = synth.grid_mesh(3, 2)
    colors = (np.arange(len(mesh.faces) * 3) % 251).astype(np.uint8).reshape(-1, 3)
    for binary in (True, False):
        p = os.path.join(tmp_path, "m_%d.ply" % binary)
        sm.data._write_ply(p, mesh.vertices, mesh.faces, colors, binary)
        back = sm.data.Ply(p)
        np.testing.assert_array_equal(back.faces, mesh.faces)
        np.testing.assert_allclose(back.vertices, mesh.vertices, rtol=1e-6)
        assert back.faces.dtype == np.int32 and back.vertices.dtype == np.float32
        with pytest.raises(ValueError):
            back.save(os.path.join(tmp_path, "x.ply"), colors[:-1])
        back.save(os.path.join(tmp_path, "y.ply"), colors)
        again = sm.data.Ply(os.path.join(tmp_path, "y.ply"))
        np.testing.assert_array_equal(again.faces, mesh.faces)


def test_grid_mesh_sizes():
    for name, tris in (("cfg1", 10_000), ("cfg2", 1_000_000)):
        cfg = synth.CONFIGS[name]
        assert 2 * cfg["a"] * cfg["b"] == tris
    m = synth.grid_mesh(4, 3)
    assert m.vertices.shape == (20, 3) and m.faces.shape == (24, 3) and m.faces.max() == 19


def test_look_at_is_a_rigid_transform():
    R, t = synth.look_at((3, 2, 5), (0, 0, 0))
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    assert abs(np.linalg.det(R) - 1) < 1e-5
    np.testing.assert_allclose(R @ np.array([3, 2, 5], np.float32) + t, 0, atol=1e-5)   # the eye maps to the origin
    assert (R @ np.zeros(3) + t)[2] > 0                                                   # the target is in front (+z)


def test_shard_views():
    for n, w in ((200, 8), (7, 3), (5, 8), (0, 2)):
        for contiguous in (True, False):
            parts = [distributed.shard_views(n, r, w, contiguous) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert distributed.shard_views(1600, 3, 8) == list(range(600, 800))
    with pytest.raises(ValueError):
        distributed.shard_views(10, 8, 8)


def _write_colmap(ws, binary):
    import struct
    cams = {1: ("PINHOLE", 640, 480, [500.5, 501.5, 320.25, 240.75]), 7: ("SIMPLE_PINHOLE", 320, 200, [250.0, 160.0, 100.0])}
    images = [(3, [0.5, 0.5, -0.5, 0.5], [0.1, -0.2, 3.0], 7, "b_second.jpg"), (9, [1, 0, 0, 0], [1, 2, 3], 1, "a_first.png")]
    ids = {"PINHOLE": 1, "SIMPLE_PINHOLE": 0}
    if binary:
        with open(os.path.join(ws, "cameras.bin"), "wb") as fh:
            fh.write(struct.pack("<Q", len(cams)))
            for cid, (m, w, h, p) in cams.items():
                fh.write(struct.pack("<iiQQ", cid, ids[m], w, h) + struct.pack("<%dd" % len(p), *p))
        with open(os.path.join(ws, "images.bin"), "wb") as fh:
            fh.write(struct.pack("<Q", len(images)))
            for iid, q, t, cid, name in images:
                fh.write(struct.pack("<I4d3dI", iid, *q, *t, cid) + name.encode() + b"\0")
                fh.write(struct.pack("<Q", 2) + struct.pack("<ddq", 1.0, 2.0, -1) * 2)
    else:
        with open(os.path.join(ws, "cameras.txt"), "w") as fh:
            fh.write("# Camera list\n")
            for cid, (m, w, h, p) in cams.items():
                fh.write("%d %s %d %d %s\n" % (cid, m, w, h, " ".join(repr(float(v)) for v in p)))
        with open(os.path.join(ws, "images.txt"), "w") as fh:
            fh.write("# Image list with two lines of data per image\n")
            for iid, q, t, cid, name in images:
                fh.write("%d %s %s %d %s\n" % (iid, " ".join(map(repr, map(float, q))), " ".join(map(repr, map(float, t))), cid, name))
                fh.write("1.0 2.0 -1 3.0 4.0 -1\n")


@pytest.mark.parametrize("binary", [True, False])
def test_colmap_reader(tmp_path, binary):
    _write_colmap(str(tmp_path), binary)
    ws = sm.data.Colmap(str(tmp_path))
    assert ws.getImageNum() == 2
    a = ws.getCamera(0)                                   # images are sorted by name (Colmap.cpp:19-21)
    assert a.resolution == (640, 480)
    np.testing.assert_allclose(a.focal_lengths, [500.5, 501.5])
    np.testing.assert_allclose(a.principal_point, [320.25, 240.75])
    np.testing.assert_allclose(a.rotation, np.eye(3), atol=1e-7)
    np.testing.assert_allclose(a.translation, [1, 2, 3])
    b = ws.getCamera("/some/dir/b_second.jpg")            # lookup by file name (Colmap.cpp:50-59)
    assert b.resolution == (320, 200) and b.focal_lengths[0] == b.focal_lengths[1] == 250.0
    np.testing.assert_allclose(b.rotation @ b.rotation.T, np.eye(3), atol=1e-6)
    np.testing.assert_allclose(b.rotation, [[0, -1, 0], [0, 0, -1], [1, 0, 0]], atol=1e-6)  # q = (w,x,y,z) = (.5,.5,-.5,.5)
    assert len(ws.getCameras()) == 2
    with pytest.raises(KeyError):
        ws.getCamera("missing.png")


def test_dlpack_roundtrip_without_gpu():
    """DeviceArray.__dlpack__ builds a well-formed capsule (kDLROCM) and describe() can consume capsules."""
    import ctypes
    from semantic_meshes_amd import dlpack
    host = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    cap = dlpack.to_capsule(host.ctypes.data, host.shape, (12, 4, 1), host.dtype, dlpack.kDLCPU, 0, host)
    n_live = len(dlpack._live)
    ptr, mem, shape, dt, strides, keep = device.describe(cap, 3, "probs")
    assert (ptr, mem, shape, dt, strides) == (host.ctypes.data, _lib.MEM_HOST, (2, 3, 4), np.dtype(np.float32), (12, 4, 1))
    keep.close()
    assert len(dlpack._live) == n_live - 1                       # the producer's deleter ran
    with pytest.raises(ValueError):
        dlpack.Imported(cap)                                     # a capsule can be consumed only once
    d = device.DeviceArray(0x4000, (5, 7), np.uint32, device=0)
    assert d.__dlpack_device__() == (dlpack.kDLROCM, 0)
    cap2 = dlpack.to_capsule(d.ptr, d.shape, d.strides, d.dtype, dlpack.kDLROCM, 0, d)
    imp = dlpack.Imported(cap2)
    assert imp.on_device and imp.ptr == 0x4000 and imp.shape == (5, 7) and imp.dtype == np.uint32 and imp.strides == (7, 1)
    imp.close()
    # numpy >= 1.22 arrays speak DLPack too: still taken through the array interface (no capsule needed)
    _, mem, *_ = device.describe(host, 3, "probs")
    assert mem == _lib.MEM_HOST


def test_hip_runtime_preload_checks_the_soname(tmp_path):
    """a bundled runtime of another ROCm major (SONAME != the DT_NEEDED of libsmesh_hip.so) must NOT be mapped --
    the linker could not reuse it and both runtimes would load."""
    import subprocess
    import sys
    soname, needed = _lib._elf_dynamic_strings(_lib.LIB_PATH)
    want = [n for n in needed if n.startswith("libamdhip64.so")]
    assert want, needed
    for tag, name in (("other", "libamdhip64.so.999"), ("same", want[0])):
        d = tmp_path / tag
        d.mkdir()
        src = d / "x.c"
        src.write_text("int smesh_fake_runtime_marker(void) { return 1; }\n")
        subprocess.check_call(["gcc", "-shared", "-fPIC", "-Wl,-soname," + name, str(src), "-o", str(d / "libamdhip64.so")])
        assert _lib._elf_dynamic_strings(str(d / "libamdhip64.so"))[0] == name
        code = ("import os, sys; sys.path.insert(0, %r); from semantic_meshes_amd import _lib; _lib._preload_hip_runtime(); "
                "print('MAPPED' if %r in open('/proc/self/maps').read() else 'LEFT')" % (os.path.dirname(os.path.dirname(__file__)), str(d)))
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SMESH_HIP_RUNTIME=str(d)),
                             capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == ("LEFT" if tag == "other" else "MAPPED"), (tag, out.stdout, out.stderr)


def test_box_extent_bound_is_a_bound():
    """smesh_box_extent_bound (raster.hip box_extent_bound: what lets the library leave out the launches for huge / clipped / big
    triangles) is plain host arithmetic in the HIP library: checked here, without a GPU, against brute force -- random soups and grids,
    random cameras outside, oblique, near and inside the mesh: wherever the function returns a finite bound, no vertex lies at or
    behind the near plane (as the float32 vertex stage computes it) and no triangle that reaches the image has a larger screen extent."""
    import ctypes
    from semantic_meshes_amd import _lib, synth, data
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("libsmesh_hip.so not built")
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.smesh_box_extent_bound.restype = ctypes.c_int
    rng = np.random.default_rng(2024)
    finite = tight = 0
    for trial in range(300):
        if trial % 3 == 0:
            mesh = synth.grid_mesh(int(rng.integers(4, 60)), int(rng.integers(4, 60)), extent=float(rng.uniform(1, 30)), relief=float(rng.uniform(0, 1)))
            V, F = np.asarray(mesh.vertices, np.float32), np.asarray(mesh.faces, np.int32)
        else:
            nv, nf = int(rng.integers(3, 200)), int(rng.integers(1, 300))
            centre = rng.uniform(-5, 5, 3)
            V = (centre + rng.normal(0, rng.uniform(0.05, 3.0), (nv, 3))).astype(np.float32)
            F = rng.integers(0, nv, (nf, 3)).astype(np.int32)
            if trial % 7 == 0:
                F[0, 1] = -1                                   # an out-of-range index: the face is ignored (load_tri)
        W, H = int(rng.integers(16, 2000)), int(rng.integers(16, 1200))
        dist = float(10 ** rng.uniform(-0.5, 3.0))
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        target = V.mean(0) + rng.normal(0, 0.5, 3)
        R, t = synth.look_at(target + d * dist, target, up=(0.3, 0.2, 1.0))
        if trial % 5 == 0:
            R = (R * np.float32(rng.uniform(0.5, 2.0))).astype(np.float32)      # not orthonormal: the caller's business
        fx, fy = float(rng.uniform(0.3, 3.0) * W), float(rng.uniform(0.3, 3.0) * W)
        cx, cy = float(rng.uniform(-0.2, 1.2) * W), float(rng.uniform(-0.2, 1.2) * H)
        cam = data.Camera(R, t, np.asarray([W, H]), np.asarray([fx, fy]), np.asarray([cx, cy]))
        bound = ctypes.c_double()
        assert L.smesh_box_extent_bound(V.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(V)), F.ctypes.data_as(ctypes.c_void_p),
                                        ctypes.c_uint64(len(F)), ctypes.byref(cam._pod), ctypes.byref(bound)) == 0
        if not np.isfinite(bound.value):
            continue
        finite += 1
        # the vertex stage: float32 rigid transform in the library's order of operations, double projection (raster.hip project_point)
        Rf, tf = np.asarray(cam._pod.rotation, np.float32).reshape(3, 3), np.asarray(cam._pod.translation, np.float32)
        Xc = np.empty_like(V)
        for r in range(3):
            Xc[:, r] = ((Rf[r, 0] * V[:, 0] + Rf[r, 1] * V[:, 1]) + Rf[r, 2] * V[:, 2]) + tf[r]
        assert (Xc[:, 2] > 1e-6).all(), "a finite bound although a vertex lies behind the near plane"
        z = Xc[:, 2].astype(np.float64)
        u = fx * (Xc[:, 0].astype(np.float64) / z) + cx
        v = fy * (Xc[:, 1].astype(np.float64) / z) + cy
        ok = (F >= 0).all(1) & (F < len(V)).all(1)
        Fi = F[ok]
        tu, tv = u[Fi], v[Fi]
        x0, x1 = np.maximum(np.ceil(tu.min(1) - 0.5), 0), np.minimum(np.floor(tu.max(1) - 0.5), W - 1)
        y0, y1 = np.maximum(np.ceil(tv.min(1) - 0.5), 0), np.minimum(np.floor(tv.max(1) - 0.5), H - 1)
        reach = (x0 <= x1) & (y0 <= y1)
        if reach.any():
            ext = max((tu.max(1) - tu.min(1))[reach].max(), (tv.max(1) - tv.min(1))[reach].max())
            assert ext <= bound.value, (trial, ext, bound.value)
            boxes = max((x1 - x0 + 1)[reach].max(), (y1 - y0 + 1)[reach].max())
            assert boxes <= bound.value + 1
            tight += ext > 0.05 * bound.value
    assert finite >= 60 and tight >= 20, (finite, tight)
