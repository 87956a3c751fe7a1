"""GPU tests at the size of the BASELINE.json configs that round 1 only ran from a dev script:

* cfg2 (1 M triangles, 1920x1080, C = 19): the entry point bench.py times -- smesh_fuse_views with a group of EIGHT views --
  against the float32 oracle, raw accumulator bit for bit;
* cfg4 (5 M triangles as texel primitives, 1296x968, C = 40): full-size properties + oracle parity on a 240 k-triangle cut;
* cfg5 (20 M triangles, 4096x2160, C = 150): full-size properties (64-bit offsets: P * C = 3e9) + oracle parity with C = 150 at
  4096x2160 on the 1 M-triangle mesh;
* the reference's only known-answer scene, python/scripts/debug_render_texels.py:19-73, verbatim.
"""
import ctypes
import math

import numpy as np
import pytest

from helpers import BG, assert_fused_close

pytestmark = pytest.mark.gpu


def _rows(agg, first, count):
    """rows [first, first + count) of the raw accumulator, without downloading the whole of it"""
    from semantic_meshes_amd import _lib
    raw = agg.raw_device_array()
    C = agg.classes
    out = np.empty((count, C), np.float32)
    _lib.check(_lib.lib().smesh_memcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(raw.ptr + first * C * 4), out.nbytes,
                                       _lib.MEM_HOST, _lib.MEM_DEVICE, agg.device))
    return out


def test_fuse_views_group_of_eight_cfg2_full_size_bit_exact(sm, oracle):
    """What bench.py times: eight views of BASELINE cfg2 per smesh_fuse_views call (grouped rasteriser launches, two views per
    fusion launch, ~240 MB of per-slot state, 64-bit row offsets at 1 M triangles), probs generated in HBM.  The float32 oracle
    consumes the same bytes one view at a time; Sum's raw accumulator must agree bit for bit."""
    from semantic_meshes_amd import synth
    mesh, cams, C = synth.scene("cfg2")
    P = len(mesh.faces)
    views = [5, 31, 57, 83, 109, 135, 161, 187]
    W, H = cams[0].resolution
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    d_probs = [synth.device_probs(W, H, C, synth.probs_seed(1, k), zero_fraction=0.03) for k in views]
    agg.fuse_views(r, [cams[k] for k in views], d_probs)
    assert sm._lib.last_fuse_kernel() == "k_fuse_tri"
    got = agg.get_raw()
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oagg = oracle.OracleAggregator(P, C)            # float32, single-threaded: the reference's own order of additions
    oracle.set_threads(8)
    try:
        oidx = [o.render(cams[k])[0] for k in views]
    finally:
        oracle.set_threads(1)
    for k, idx, dp in zip(views, oidx, d_probs):
        np.testing.assert_array_equal(np.asarray(r.render(cams[k])[0]), idx)
        oagg.add(idx, np.asarray(dp))
    want = oagg.get_raw()
    assert (np.abs(want).sum(axis=1) > 0).sum() > 900_000
    np.testing.assert_array_equal(got.view(np.uint32), want.view(np.uint32))
    np.testing.assert_array_equal(agg.get().view(np.uint32), oagg.get().view(np.uint32))


def test_known_answer_scene_of_the_reference(sm, oracle):
    """python/scripts/debug_render_texels.py:19-73, verbatim: the triangle (0.4,0,0), (0.5,1,0), (0.6,0,0) in all six vertex
    orders, look-at camera at 4000 x 4000 with a 45 degree field of view, texels(mesh, [camera], 0.01).  The script prints
    `max(idx) + 1` texels and the side length `int(-0.5 + sqrt(0.25 + 2 n))`: n must be r (r + 1) / 2; the coverage and the
    depth must not depend on the vertex order; and the HIP path must equal the oracle bit for bit at this size."""
    vertices = np.array([(0.4, 0, 0), (0.5, 1, 0), (0.6, 0, 0)], np.float32)
    # pyrr.matrix44.create_look_at(eye, target, up), transposed and inverted as the script does (:46-54)
    eye, target, up = np.array([-0.5, -0.5, 4.0]), np.array([-0.5, -0.5, 0.0]), np.array([0.0, 1.0, 0.0])
    fwd = (target - eye) / np.linalg.norm(target - eye)
    side = np.cross(fwd, up) / np.linalg.norm(np.cross(fwd, up))
    upv = np.cross(side, fwd)
    look_at = np.array([[side[0], upv[0], -fwd[0], 0], [side[1], upv[1], -fwd[1], 0], [side[2], upv[2], -fwd[2], 0],
                        [-side @ eye, -upv @ eye, fwd @ eye, 1]], np.float32)
    camera_to_world = np.linalg.inv(np.transpose(look_at, (1, 0)))
    rotation, translation = camera_to_world[:3, :3], camera_to_world[:3, 3]
    resolution = np.asarray([4000, 4000])
    principal_point = resolution.astype("float32") / 2.0
    fov_y = math.radians(45.0)
    focal_lengths = np.asarray([principal_point[0] / (resolution[0] / resolution[1] * math.tan(fov_y / 2.0)),
                                principal_point[1] / math.tan(fov_y / 2.0)])
    camera = sm.data.Camera(rotation, translation, np.asarray([resolution[1], resolution[0]]), focal_lengths, principal_point)
    coverage, depths, counts = [], [], []
    for face in [[0, 1, 2], [0, 2, 1], [1, 0, 2], [1, 2, 0], [2, 0, 1], [2, 1, 0]]:
        mesh = sm.data.Mesh(vertices, np.array([face], np.int32))
        renderer = sm.render.texels(mesh, [camera], 0.01)
        primitive_indices, depth = renderer.render(camera)
        idx, dep = np.asarray(primitive_indices), np.asarray(depth)
        o = oracle.OracleRenderer(vertices, np.array([face], np.int32), [camera], 0.01)
        oidx, odep = o.render(camera)
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(dep.view(np.uint32), odep.view(np.uint32))
        _, res, first = renderer.texel_layout()
        r = int(res[0])
        classes_num = int(idx[idx != BG].max()) + 1                       # :70
        sidelength = int(-0.5 + math.sqrt(0.25 + 2 * classes_num))        # :71
        assert classes_num == renderer.getPrimitivesNum() == r * (r + 1) // 2 and sidelength == r and r >= 3
        assert len(np.unique(idx[idx != BG])) == classes_num              # every texel of the triangle is seen
        assert np.isinf(dep[idx == BG]).all() and np.isfinite(dep[idx != BG]).all()
        coverage.append(idx != BG)
        depths.append(dep)
        counts.append(np.sort(np.bincount(idx[idx != BG])))
    assert 100_000 < coverage[0].sum() < 200_000                          # ~0.5 * 241 * 1207 pixels
    for c, d, n in zip(coverage[1:], depths[1:], counts[1:]):
        np.testing.assert_array_equal(c, coverage[0])                     # same pixels whatever the vertex order ...
        np.testing.assert_allclose(d[c], depths[0][c], rtol=1e-6)         # ... at the same depth (a plane: z = 4) ...
        np.testing.assert_allclose(n, counts[0], rtol=0.02, atol=40)      # ... cut into texels of the same sizes


def test_cfg4_texels_parity_on_a_240k_triangle_cut(sm, oracle):
    """BASELINE cfg4's path (render.texels + fuse, 1296x968, C = 40) at a size the oracle affords: 240 k triangles, texel
    resolutions up to 3 (texels_per_pixel 0.6), three cameras.  Layout, index and depth images bit-exact; Sum's raw accumulator
    against the float32 oracle to 2e-6, get() within 1e-5."""
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg4"]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(400, 300)
    cams = [synth.ring_camera(k, 7, W, H) for k in (0, 2, 5)]
    r = sm.render.texels(mesh, cams, 0.6)
    oracle.set_threads(8)
    try:
        o = oracle.OracleRenderer(mesh.vertices, mesh.faces, cams, 0.6)
        faces, res, first = r.texel_layout()
        ofaces, ores, ofirst = o.texel_layout()
        np.testing.assert_array_equal(faces, ofaces)
        np.testing.assert_array_equal(res, ores)
        np.testing.assert_array_equal(first, ofirst)
        P = r.getPrimitivesNum()
        assert P == o.getPrimitivesNum() and P > 2 * len(mesh.faces) and res.max() >= 3
        oidx = []
        for cam in cams:
            idx, depth = r.render(cam)
            oi, od = o.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oi)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), od.view(np.uint32))
            oidx.append(oi)
    finally:
        oracle.set_threads(1)
    agg = sm.fusion.MeshAggregator(P, C)
    oagg = oracle.OracleAggregator(P, C)
    for k, cam in enumerate(cams):
        dp = synth.device_probs(W, H, C, synth.probs_seed(4, k), zero_fraction=0.03)
        agg.fuse_view(r, cam, dp)
        oagg.add(oidx[k], np.asarray(dp))
    assert sm._lib.last_fuse_kernel() == "k_fuse_texel"
    # (not bit for bit: triangles with a box over 8 x 8 pixels are fused by a whole wave with float atomics on their own rows)
    np.testing.assert_allclose(agg.get_raw(), oagg.get_raw(), rtol=2e-6, atol=1e-7)
    assert_fused_close(agg.get(), oagg.get())


def test_cfg4_full_size_properties(sm):
    """BASELINE cfg4 at full size: 5 M triangles as texel primitives, 1296x968, C = 40 (size-independent properties)."""
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg4"]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    assert len(mesh.faces) == 5_000_000
    ctor_cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(0, cfg["views"], 50)]
    r = sm.render.texels(mesh, ctor_cams, 0.1)
    P = r.getPrimitivesNum()
    _, res, first = r.texel_layout()
    assert P == int((res.astype(np.int64) * (res + 1) // 2).sum()) >= (res > 0).sum() > 4_000_000       # KA10
    assert (np.diff(first.astype(np.int64)) == (res[:-1].astype(np.int64) * (res[:-1] + 1) // 2)).all()
    views = [3, 400, 777]
    cams = [synth.ring_camera(k, cfg["views"], W, H) for k in views]
    probs = [synth.device_probs(W, H, C, synth.probs_seed(4, k), 0.02) for k in views]
    idx_a = np.asarray(r.render(cams[0])[0])
    assert np.array_equal(idx_a, np.asarray(r.render(cams[0])[0]))                               # idempotent
    valid = idx_a[idx_a != BG]
    assert valid.max() < P and len(np.unique(valid)) > 300_000 and (idx_a != BG).mean() > 0.4
    whole = sm.fusion.MeshAggregator(P, C)
    parts = [sm.fusion.MeshAggregator(P, C) for _ in views]
    whole.fuse_views(r, cams, probs)                                                             # grouped rasteriser launches
    assert sm._lib.last_fuse_kernel() == "k_fuse_texel"
    for cam, p, part in zip(cams, probs, parts):
        part.fuse_view(r, cam, p)
    raw_sum = sum(part.get_raw().astype(np.float64) for part in parts)
    np.testing.assert_allclose(whole.get_raw(), raw_sum, rtol=1e-5, atol=1e-6)                   # shard-sum identity
    out = whole.get()
    touched = out.sum(axis=1) > 0.5
    np.testing.assert_allclose(out[touched].sum(axis=1), 1.0, rtol=1e-5)                         # rows L1-normalised
    assert (out[~touched] == 0).all() and touched.sum() > 1_000_000
    plain = sm.fusion.MeshAggregator(P, C, "sum", 0.0)       # mass conservation with iew = 0
    plain.fuse_view(r, cams[0], probs[0])
    hp = np.asarray(probs[0]).reshape(-1, C)
    keep = (idx_a.reshape(-1) != BG) & (hp.sum(axis=1) > 0.5)
    np.testing.assert_allclose(plain.get_raw().astype(np.float64).sum(), hp[keep].astype(np.float64).sum(), rtol=1e-5)


def test_cfg4t_multi_texel_full_size_properties(sm):
    """cfg4's mesh with texels_per_pixel 5 (`bench.py --workload cfg4t`): the sub-pixel triangles get three texel rows
    (six texels each), so the texel shader runs with r > 1 at full size.  Size-independent properties."""
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg4t"]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    ctor_cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(0, cfg["views"], 50)]
    r = sm.render.texels(mesh, ctor_cams, cfg["texels_per_pixel"])
    P = r.getPrimitivesNum()
    _, res, first = r.texel_layout()
    assert np.median(res) >= 3 and P == int((res.astype(np.int64) * (res + 1) // 2).sum()) > 25_000_000          # KA10
    views = [3, 400, 777]
    cams = [synth.ring_camera(k, cfg["views"], W, H) for k in views]
    probs = [synth.device_probs(W, H, C, synth.probs_seed(4, k), 0.02) for k in views]
    idx_a = np.asarray(r.render(cams[0])[0])
    assert np.array_equal(idx_a, np.asarray(r.render(cams[0])[0]))                               # idempotent
    valid = idx_a[idx_a != BG]
    assert valid.max() < P and (idx_a != BG).mean() > 0.4
    # every visible texel belongs to a triangle with texels, and all of a triangle's texel slots are hit somewhere in the image
    tri = np.searchsorted(first.astype(np.int64), valid.astype(np.int64), side="right") - 1
    local = valid.astype(np.int64) - first[tri].astype(np.int64)
    assert (local < res[tri].astype(np.int64) * (res[tri] + 1) // 2).all()
    assert len(np.unique(local)) >= 6                                                           # r >= 3: texels 0 .. 5 all occur
    whole = sm.fusion.MeshAggregator(P, C)
    parts = [sm.fusion.MeshAggregator(P, C) for _ in views]
    whole.fuse_views(r, cams, probs)
    assert sm._lib.last_fuse_kernel() == "k_fuse_texel"
    for cam, p, part in zip(cams, probs, parts):
        part.fuse_view(r, cam, p)
    # shard-sum identity on every fifth block of a million rows (the whole 4.8 GB accumulator in float64 is 10 GB of host memory)
    raw = whole.get_raw()
    part_raw = [part.get_raw() for part in parts]
    del parts
    for lo in range(0, P, 5_000_000):
        sl = slice(lo, min(lo + 1_000_000, P))
        np.testing.assert_allclose(raw[sl], sum(p[sl].astype(np.float64) for p in part_raw), rtol=1e-5, atol=1e-6)
    assert (raw.sum(axis=1) > 0).sum() > 1_000_000
    del raw, part_raw
    plain = sm.fusion.MeshAggregator(P, C, "sum", 0.0)       # mass conservation with iew = 0
    plain.fuse_view(r, cams[0], probs[0])
    hp = np.asarray(probs[0]).reshape(-1, C)
    keep = (idx_a.reshape(-1) != BG) & (hp.sum(axis=1) > 0.5)
    np.testing.assert_allclose(plain.get_raw().astype(np.float64).sum(), hp[keep].astype(np.float64).sum(), rtol=1e-5)


def test_cfg5_class_count_and_resolution_parity_on_the_1m_triangle_mesh(sm, oracle):
    """BASELINE cfg5's fusion kernel (k_fuse_tri_wide, C = 150) at cfg5's resolution (4096x2160, 5.3 GB of class vectors per
    view) on the 1 M-triangle mesh, which the oracle affords: indices and depth bit-exact, Sum's raw accumulator to 2e-6."""
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg5"]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(1000, 500)
    P = len(mesh.faces)
    cam = synth.ring_camera(17, 200, W, H)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    dp = synth.device_probs(W, H, C, synth.probs_seed(5, 17), zero_fraction=0.02)
    agg.fuse_view(r, cam, dp)
    assert sm._lib.last_fuse_kernel() == "k_fuse_tri_wide"
    oracle.set_threads(8)
    try:
        oidx, odepth = oracle.OracleRenderer(mesh.vertices, mesh.faces).render(cam)
    finally:
        oracle.set_threads(1)
    idx, depth = r.render(cam)
    np.testing.assert_array_equal(np.asarray(idx), oidx)
    np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
    oagg = oracle.OracleAggregator(P, C)
    oagg.add(oidx, np.asarray(dp))
    # (not bit for bit: at this resolution many triangles of the 1 M mesh have a box over 8 x 8 pixels and are summed by a whole
    # wave in tree order; small-triangle scenes with C = 150 are bit-equal, test_fuse_view_triangle_order_is_bit_exact)
    np.testing.assert_allclose(agg.get_raw(), oagg.get_raw(), rtol=2e-6, atol=1e-7)
    assert_fused_close(agg.get(), oagg.get())


def test_cfg5_full_size_properties(sm):
    """BASELINE cfg5 at full size: 20 M triangles, 4096x2160, C = 150 -- a 12 GB accumulator (P * C = 3e9 floats: row offsets
    beyond 32 bits), 5.3 GB of class vectors per view.  Checked through windows of rows (the whole accumulator is not downloaded)."""
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg5"]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    P = len(mesh.faces)
    assert P == 20_000_000 and P * C > 2 ** 31
    r = sm.render.triangles(mesh)
    views = [40, 290]
    cams = [synth.ring_camera(k, cfg["views"], W, H) for k in views]
    idx = [np.asarray(r.render(cam)[0]) for cam in cams]
    assert np.array_equal(idx[0], np.asarray(r.render(cams[0])[0]))                              # idempotent
    for im in idx:
        valid = im[im != BG]
        assert valid.max() < P and (im != BG).mean() > 0.4 and len(np.unique(valid)) > 2_000_000
    assert max(int(im[im != BG].max()) for im in idx) * C * 4 > 2 ** 33                          # rows past the 8 GiB mark are hit
    probs = synth.device_probs(W, H, C, synth.probs_seed(5, 1), 0.02)                            # one 5.3 GB image for both views
    whole = sm.fusion.MeshAggregator(P, C)
    parts = [sm.fusion.MeshAggregator(P, C) for _ in views]
    whole.fuse_views(r, cams, [probs, probs])
    assert sm._lib.last_fuse_kernel() == "k_fuse_tri_wide"
    for cam, part in zip(cams, parts):
        part.fuse_view(r, cam, probs)
    hp = None
    for im in idx:                                     # windows of 200 k rows around the lowest / median / highest visible primitive
        valid = np.unique(im[im != BG])
        for centre in (int(valid[0]), int(valid[len(valid) // 2]), int(valid[-1])):
            first = max(0, min(P - 200_000, centre - 100_000))
            got = _rows(whole, first, 200_000).astype(np.float64)
            want = sum(_rows(part, first, 200_000).astype(np.float64) for part in parts)
            np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)                          # shard-sum identity
            assert (np.abs(got).sum(axis=1) > 0).sum() > 1000
    # one view, iew = 0: every visible pixel's class vector lands in its primitive's row -- checked on the rows of one window
    del whole, parts
    plain = sm.fusion.MeshAggregator(P, C, "sum", 0.0)
    plain.fuse_view(r, cams[0], probs)
    valid = np.unique(idx[0][idx[0] != BG])
    first = max(0, min(P - 100_000, int(valid[-1]) - 99_999))
    rows = _rows(plain, first, 100_000).astype(np.float64)
    flat = idx[0].reshape(-1)
    sel = np.flatnonzero((flat >= first) & (flat < first + 100_000))
    hp = np.empty((len(sel), C), np.float32)
    from semantic_meshes_amd import _lib
    # gather the selected pixels' class vectors from HBM in runs (the image is 5.3 GB)
    runs = np.split(sel, np.flatnonzero(np.diff(sel) != 1) + 1)
    at = 0
    for run in runs:
        n = len(run)
        _lib.check(_lib.lib().smesh_memcpy(hp[at:at + n].ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(probs.ptr + int(run[0]) * C * 4),
                                           n * C * 4, _lib.MEM_HOST, _lib.MEM_DEVICE, 0))
        at += n
    keep = hp.sum(axis=1) > 0.5
    want = np.zeros((100_000, C), np.float64)
    np.add.at(want, flat[sel][keep] - first, hp[keep].astype(np.float64))
    np.testing.assert_allclose(rows, want, rtol=1e-5, atol=1e-6)
    out = plain.get_device()                                                                     # normalised, stays in HBM
    tail = np.empty((100_000, C), np.float32)
    _lib.check(_lib.lib().smesh_memcpy(tail.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(out.ptr + first * C * 4), tail.nbytes,
                                       _lib.MEM_HOST, _lib.MEM_DEVICE, 0))
    touched = want.sum(axis=1) > 0.5
    np.testing.assert_allclose(tail[touched].sum(axis=1), 1.0, rtol=1e-5)                        # rows L1-normalised
    assert (tail[~touched] == 0).all() and touched.sum() > 1000


def test_group_of_eight_cfg2_with_three_groups_of_triangles_per_wave():
    """SMESH_RASTER_GROUPS=3 (read once per process): k_raster_frag_group's waves take three groups of 64 triangles each, the next
    group's vertex indices prefetched -- the arrangement meshes of four million triangles and more get by default -- on the
    full-size cfg2 test above, whose 1 M triangles are not a multiple of 192."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SMESH_RASTER_GROUPS="3")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_configs.py"), "-q", "-x", "-m", "gpu", "-k",
                          "group_of_eight_cfg2_full_size", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
