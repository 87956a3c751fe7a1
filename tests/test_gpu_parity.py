"""GPU parity tests: HIP path (through the C ABI via ctypes) vs the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): primitive indices bit-exact, fused float32 distributions within 1e-5 relative.
"""
import numpy as np
import pytest

from helpers import BG, assert_fused_close, random_probs, small_scene

pytestmark = pytest.mark.gpu


def _render_both(sm, oracle, mesh, cam):
    r = sm.render.triangles(mesh)
    idx, depth = r.render(cam)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oidx, odepth = o.render(cam)
    return np.asarray(idx), np.asarray(depth), oidx, odepth


def test_backend_is_hip(sm):
    from semantic_meshes_amd import _lib
    assert _lib.lib().smesh_backend() == b"hip-gfx950"


@pytest.mark.parametrize("view", [0, 1, 2])
def test_render_small_scene_bit_exact(sm, oracle, view):
    mesh, cams = small_scene()
    idx, depth, oidx, odepth = _render_both(sm, oracle, mesh, cams[view])
    assert idx.dtype == np.uint32 and idx.shape == cams[view].resolution
    assert (idx != BG).mean() > 0.3
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(depth.view(np.uint32), odepth.view(np.uint32))


def test_render_cfg1_bit_exact(sm, oracle):
    from semantic_meshes_amd import synth
    mesh, cams, _ = synth.scene("cfg1")
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    assert r.getPrimitivesNum() == len(mesh.faces) == 10000
    for cam in cams:
        idx, depth = r.render(cam)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))


def test_render_cfg2_full_size_bit_exact(sm, oracle):
    """BASELINE cfg2: 1M triangles at 1920x1080 -- the oracle renders a view in ~0.2 s."""
    from semantic_meshes_amd import synth
    mesh, cams, _ = synth.scene("cfg2")
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    for k in (0, 66, 133):
        idx, depth = r.render(cams[k])
        oidx, odepth = o.render(cams[k])
        idx = np.asarray(idx)
        assert (idx != BG).mean() > 0.5
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))


def test_near_plane_clipping_matches_oracle(sm, oracle):
    """Raster spec B-3 (round 3): triangles that cross the near plane z_c = 1e-6 are clipped against it
    (tests/test_oracle.py pins the rule on the oracle and against an independent ray caster; here the HIP path agrees bit for bit)."""
    from test_oracle import near_plane_scene
    cam, v, f = near_plane_scene()
    for z in (-1.0, 0.0, 1e-6, 2e-6, 0.5):
        vv = v.copy()
        vv[5, 2] = z
        idx, depth = sm.render.triangles(sm.data.Mesh(vv, f)).render(cam)
        oidx, odepth = oracle.OracleRenderer(vv, f).render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
        assert (oidx == 1).sum() > 500


@pytest.mark.parametrize("res", [(96, 72), (640, 480), (1296, 968)])
def test_camera_inside_a_room_no_holes_and_bit_equal(sm, oracle, res):
    """the camera inside a closed box (a ScanNet room seen from within, eval-scannet/eval_scannet.py:203-238) -- the
    walls that cross the camera plane are visible, no pixel sees the background, indices and depth bit-equal to the oracle;
    the class vectors fused on them as well (they take the big-triangle path of the fusion)."""
    from test_oracle import room_scene
    W, H = res
    poses = [((0.3, -0.2, 0.1), (2.0, 0.5, 0.0)), ((-1.2, 0.9, -0.6), (0.0, -1.5, 0.4)), ((1.5, 1.2, 1.0), (-2.0, -1.5, -1.2))]
    cams = [room_scene(W=W, H=H, eye=e, target=t, f=0.55 * W)[0] for e, t in poses]
    _, v, f = room_scene()
    r = sm.render.triangles(sm.data.Mesh(v, f))
    o = oracle.OracleRenderer(v, f)
    C = 7
    agg = sm.fusion.MeshAggregator(len(f), C)
    agg2 = sm.fusion.MeshAggregator(len(f), C)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(len(f), C)
        probs = []
        for k, cam in enumerate(cams):
            idx, depth = r.render(cam)
            oidx, odepth = o.render(cam)
            assert (oidx != BG).all()
            np.testing.assert_array_equal(np.asarray(idx), oidx)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
            p = oracle.synth_probs(W * H, C, 77 + k, 0.05).reshape(W, H, C)
            probs.append(p)
            agg.add(idx, p)
            oagg.add(oidx, p)
        from semantic_meshes_amd.device import to_device
        agg2.fuse_views(r, cams, [to_device(p) for p in probs])
        want = oagg.get()
    finally:
        oracle.set_accum_double(False)
    assert_fused_close(agg.get(), want)
    assert_fused_close(agg2.get(), want)


def test_room_with_texel_primitives_bit_equal(sm, oracle):
    from test_oracle import room_scene, _camera
    cam, v, f = room_scene(W=320, H=240, f=160.0)
    ctor = [_camera(W=320, H=240, eye=(0.0, 0.0, 9.0), target=(0, 0, 0), f=100.0), cam]
    r = sm.render.texels(sm.data.Mesh(v, f), ctor, 0.2)
    o = oracle.OracleRenderer(v, f, cameras=ctor, texels_per_pixel=0.2)
    assert r.getPrimitivesNum() == o.getPrimitivesNum() > 40
    for c in (cam, room_scene(W=320, H=240, eye=(-1.2, 0.9, -0.6), target=(0.0, -1.5, 0.4), f=160.0)[0]):
        idx, depth = r.render(c)
        oidx, odepth = o.render(c)
        assert (oidx != BG).mean() > 0.9
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))


def test_render_is_deterministic(sm):
    mesh, cams = small_scene(60, 30, 320, 240)
    r = sm.render.triangles(mesh)
    a = np.asarray(r.render(cams[0])[0])
    for _ in range(3):
        np.testing.assert_array_equal(np.asarray(r.render(cams[0])[0]), a)


def test_render_empty_scene_is_background(sm):
    # KA1: TriangleRenderer.h:75-78
    mesh = sm.data.Mesh(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    from semantic_meshes_amd import synth
    cam = synth.ring_camera(0, 1, 64, 48)
    idx, depth = sm.render.triangles(mesh).render(cam)
    assert (np.asarray(idx) == BG).all() and np.isposinf(np.asarray(depth)).all()


def test_render_overlap_nearest_wins_and_big_triangles(sm, oracle):
    # KA2/KA3: two large camera-facing triangles at different depths (cooperative big-triangle path)
    from semantic_meshes_amd import synth
    v = np.array([[-4, -3, 0], [4, -3, 0], [0, 4, 0], [-2, -3, 1], [6, -3, 1], [2, 4, 1]], np.float32)
    f = np.array([[0, 1, 2], [3, 5, 4]], np.int32)
    mesh = sm.data.Mesh(v, f)
    R, t = synth.look_at((0.5, 0.2, 8.0), (0, 0, 0), up=(0, 1, 0))
    cam = sm.data.Camera(R, t, np.array([200, 150]), np.array([160.0, 160.0]), np.array([100.0, 75.0]))
    idx, depth, oidx, odepth = _render_both(sm, oracle, mesh, cam)
    np.testing.assert_array_equal(idx, oidx)
    np.testing.assert_array_equal(depth.view(np.uint32), odepth.view(np.uint32))
    covered = idx[idx != BG]
    assert covered.size > 2000 and set(np.unique(covered)) == {0, 1}
    both = (oidx == 1).sum()
    assert both > 1000 and (oidx == 0).sum() > 1000
    # where the two overlap the nearer one (world z = 1, camera at z = 8) wins: KA3
    assert idx[110, 80] == 1 and np.isclose(depth[110, 80], 7.0, atol=0.2)


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("iew", [0.5, 0.0, 1.0])
def test_fusion_small_matches_oracle(sm, oracle, kind, iew):
    mesh, cams = small_scene()
    C = 7
    rng = np.random.default_rng(5)
    P = len(mesh.faces)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(primitives=P, classes=C, aggregator=kind, images_equal_weight=iew)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind, iew)
        for cam in cams:
            idx, _ = r.render(cam)
            probs = random_probs(rng, *cam.resolution, C)
            if kind == "mul":
                probs = np.maximum(probs, 1e-3).astype(np.float32)  # keep log p finite for a meaningful comparison
            agg.add(idx, probs)
            oagg.add(np.asarray(idx), probs)
        got, want = agg.get(), oagg.get()
    finally:
        oracle.set_accum_double(False)
    assert got.dtype == np.float32 and got.shape == (P, C)
    # Mul too: one formula on both sides down to the logarithm (a fixed float32 operation sequence), (hi, lo) state folded in
    # double
    assert_fused_close(got, want, rtol=1e-5)
    touched = want.sum(axis=1) > 0.5
    assert touched.sum() > P // 4
    np.testing.assert_allclose(got[touched].sum(axis=1), 1.0, rtol=1e-5)   # KA8
    if kind != "mul":
        assert (got[~touched] == 0).all()                                    # KA7


@pytest.mark.parametrize("C", [1, 5, 19, 40, 64, 150, 256])
def test_fusion_class_counts(sm, oracle, C):
    rng = np.random.default_rng(C)
    W, H, P = 96, 70, 500
    idx = rng.integers(0, P + 40, size=(W, H)).astype(np.uint32)
    idx[rng.random((W, H)) < 0.2] = BG
    idx = np.sort(idx.reshape(-1)).reshape(W, H)  # long runs and short runs
    idx[::7] = rng.integers(0, P, size=idx[::7].shape)
    probs = random_probs(rng, W, H, C)
    weights = rng.random((W, H), dtype=np.float32)
    agg = sm.fusion.MeshAggregator(P, C)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for _ in range(2):
            agg.add(idx, probs, weights)
            oagg.add(idx, probs, weights)
        assert_fused_close(agg.get(), oagg.get())
        assert_fused_close(agg.get_raw(), oagg.get_raw(), rtol=1e-5, atol=1e-7)
    finally:
        oracle.set_accum_double(False)


def test_fusion_dtypes_strides_and_device_inputs(sm, oracle):
    from semantic_meshes_amd.device import to_device
    rng = np.random.default_rng(11)
    W, H, C, P = 64, 48, 19, 300
    idx = rng.integers(0, P, size=(W, H)).astype(np.int64)
    idx[rng.random((W, H)) < 0.1] = -1                       # KA4: -1 of any int type is background
    hwc = random_probs(rng, H, W, C)                          # network layout (H,W,C)
    probs = hwc.transpose(1, 0, 2)                            # callers transpose without copying
    want_agg = oracle.OracleAggregator(P, C)
    want_agg.add(idx, np.ascontiguousarray(probs))
    want = want_agg.get()
    for dt in (np.int64, np.uint64, np.int32, np.uint32):
        agg = sm.fusion.MeshAggregator(P, C)
        agg.add(idx.astype(dt), probs)
        assert_fused_close(agg.get(), want)
    # device-resident inputs, transposed view on the device, Fortran-ordered index image
    agg = sm.fusion.MeshAggregator(P, C)
    d_hwc = to_device(hwc)
    d_idx = to_device(np.ascontiguousarray(idx.astype(np.int32).T))   # stored (H,W)
    agg.add(d_idx.transpose(1, 0), d_hwc.transpose(1, 0, 2))
    assert_fused_close(agg.get(), want)


def test_fusion_dont_care_pixels_still_counted(sm, oracle):
    # KA5 (Mesh.h:90-93 counts every pixel; :98 filters later) and KA6 (iew = 1: each image weighs 1 per primitive)
    W, H, C, P = 8, 8, 3, 2
    idx = np.zeros((W, H), np.uint32)
    probs = np.zeros((W, H, C), np.float32)
    probs[:4, :, 0] = 1.0                                     # half of the pixels are don't-care (sum 0)
    agg = sm.fusion.MeshAggregator(P, C, "sum", 1.0)
    agg.add(idx, probs)
    raw = agg.get_raw()
    np.testing.assert_allclose(raw[0], [0.5, 0, 0], rtol=1e-6)  # 32 valid pixels * (1/64)
    assert (raw[1] == 0).all()
    oagg = oracle.OracleAggregator(P, C, "sum", 1.0)
    oagg.add(idx, probs)
    np.testing.assert_allclose(oagg.get_raw(), raw, rtol=1e-6)


def test_fusion_reset_and_raw_roundtrip(sm):
    rng = np.random.default_rng(3)
    W, H, C, P = 32, 32, 5, 50
    idx = rng.integers(0, P, size=(W, H)).astype(np.uint32)
    probs = random_probs(rng, W, H, C, 0.0)
    agg = sm.fusion.MeshAggregator(P, C)
    agg.add(idx, probs)
    raw = agg.get_raw()
    first = agg.get()
    agg.reset()
    assert (agg.get_raw() == 0).all() and (agg.get() == 0).all()
    agg.set_raw(raw)
    np.testing.assert_array_equal(agg.get(), first)
    # sharding equivalence on one GPU: sum of two half-jobs' raw buffers == the whole job
    a1, a2 = sm.fusion.MeshAggregator(P, C), sm.fusion.MeshAggregator(P, C)
    a1.add(idx, probs)
    a2.add(idx[::-1].copy(), probs)
    whole = sm.fusion.MeshAggregator(P, C)
    whole.add(idx, probs)
    whole.add(idx[::-1].copy(), probs)
    a1.set_raw(a1.get_raw() + a2.get_raw())
    assert_fused_close(a1.get(), whole.get())


def test_fusion_errors(sm):
    agg = sm.fusion.MeshAggregator(10, 4)
    idx = np.zeros((8, 6), np.uint32)
    with pytest.raises(ValueError):
        agg.add(idx, np.zeros((8, 7, 4), np.float32))          # KA9: Mesh.h:68-74
    with pytest.raises(ValueError):
        agg.add(idx, np.zeros((8, 6, 5), np.float32))          # wrong class count
    with pytest.raises(ValueError):
        agg.add(idx.astype(np.float32), np.zeros((8, 6, 4), np.float32))
    with pytest.raises(ValueError):
        agg.add(idx, np.zeros((8, 6, 4), np.float32), np.zeros((6, 8), np.float32))
    with pytest.raises(ValueError):
        sm.fusion.MeshAggregator(10, 4, aggregator="median")


def test_fuse_view_cfg2_full_size(sm, oracle):
    """One full-size BASELINE cfg2 view through the fused entry point with probs generated in HBM;
    the oracle consumes the very same bytes (downloaded)."""
    from semantic_meshes_amd import synth
    mesh, cams, C = synth.scene("cfg2")
    P = len(mesh.faces)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for k in (3, 120):
            W, H = cams[k].resolution
            d_probs = synth.device_probs(W, H, C, synth.probs_seed(1, k), zero_fraction=0.05)
            agg.fuse_view(r, cams[k], d_probs)
            h_probs = np.asarray(d_probs)
            np.testing.assert_array_equal(h_probs.reshape(-1, C), oracle.synth_probs(W * H, C, synth.probs_seed(1, k), 0.05))
            oagg.add(o.render(cams[k])[0], h_probs)
        assert_fused_close(agg.get(), oagg.get())
    finally:
        oracle.set_accum_double(False)


def test_golden_fixtures_on_gpu(sm):
    """The committed oracle fixtures (tests/golden/, generated by make_golden.py) against the HIP path."""
    import os
    from semantic_meshes_amd import synth
    from oracle import oracle as o
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gdir, "cfg1_render.npz"))
    want_fuse = np.load(os.path.join(gdir, "cfg1_fuse.npz"))
    mesh = sm.data.Mesh(g["vertices"], g["faces"])
    r = sm.render.triangles(mesh)
    aggs = {k: sm.fusion.MeshAggregator(10000, 5, k, 0.5) for k in ("sum", "summax", "mul")}
    for k in range(4):
        cam = sm.data.Camera(g["cam%d_R" % k], g["cam%d_t" % k], g["cam%d_res" % k], g["cam%d_f" % k], g["cam%d_c" % k])
        idx, depth = r.render(cam)
        want = np.repeat(g["idx%d_vals" % k], g["idx%d_lens" % k]).reshape(cam.resolution)
        np.testing.assert_array_equal(np.asarray(idx), want)
        assert np.bitwise_xor.reduce(np.asarray(depth).view(np.uint32).reshape(-1)) == g["depth%d_xor" % k]
        W, H = cam.resolution
        probs = o.synth_probs(W * H, 5, synth.probs_seed(7, k), 0.05).reshape(W, H, 5)
        for a in aggs.values():
            a.add(idx, probs)
    for kind, a in aggs.items():
        # Sum / Summax: the fixture was accumulated in float32, single-threaded, like the reference -- and so does the HIP path: equal bits.
        # Mul: the fixture of the float64-accumulating yardstick (the float32 log-domain state the reference keeps is itself 1e-4 off
        # after four views: tests/test_oracle.py checks that fixture against this one)
        # Sum / Summax: the fixture was accumulated in float32 in pixel order; cfg1's thirty-pixel triangles are summed by a wave each (a tree)
        assert_fused_close(a.get(), want_fuse["mul_float64_state" if kind == "mul" else kind], rtol=1e-5)


def test_texel_renderer_matches_oracle(sm, oracle):
    mesh, cams = small_scene(24, 12, 200, 150, views=4)
    r = sm.render.texels(mesh, cams, 0.4)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces, cams, 0.4)
    faces, res, first = r.texel_layout()
    ofaces, ores, ofirst = o.texel_layout()
    np.testing.assert_array_equal(faces, ofaces)                     # vertex re-ordering (TexturedTriangleRenderer.h:129-146)
    np.testing.assert_array_equal(res, ores)
    np.testing.assert_array_equal(first, ofirst)
    assert r.getPrimitivesNum() == o.getPrimitivesNum() == int((res.astype(np.int64) * (res + 1) // 2).sum())   # KA10
    assert r.getPrimitivesNum() > len(mesh.faces)
    for cam in cams:
        idx, depth = r.render(cam)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
    # texel primitives through the aggregator
    P, C = r.getPrimitivesNum(), 4
    rng = np.random.default_rng(2)
    agg, oagg = sm.fusion.MeshAggregator(P, C), oracle.OracleAggregator(P, C)
    probs = random_probs(rng, *cams[0].resolution, C)
    idx, _ = r.render(cams[0])
    agg.add(idx, probs)
    oagg.add(np.asarray(idx), probs)
    assert_fused_close(agg.get(), oagg.get())


def test_full_size_properties_cfg2(sm):
    """Size-independent properties at BASELINE cfg2 scale: idempotent render, linear accumulation,
    and shard-sum == whole job (the all-reduce identity) on one GPU."""
    from semantic_meshes_amd import synth
    mesh, cams, C = synth.scene("cfg2")
    P = len(mesh.faces)
    r = sm.render.triangles(mesh)
    W, H = cams[0].resolution
    views = [10, 90, 170]
    probs = [synth.device_probs(W, H, C, synth.probs_seed(2, k), 0.02) for k in views]
    idx_a = np.asarray(r.render(cams[10])[0])
    assert np.array_equal(idx_a, np.asarray(r.render(cams[10])[0]))                      # idempotent
    valid = idx_a[idx_a != BG]
    assert valid.max() < P and len(np.unique(valid)) > 400_000
    whole = sm.fusion.MeshAggregator(P, C)
    parts = [sm.fusion.MeshAggregator(P, C) for _ in views]
    for v, p, part in zip(views, probs, parts):
        whole.fuse_view(r, cams[v], p)
        part.fuse_view(r, cams[v], p)
    raw_sum = sum(part.get_raw().astype(np.float64) for part in parts)
    np.testing.assert_allclose(whole.get_raw(), raw_sum, rtol=1e-5, atol=1e-6)           # shard-sum identity
    twice = sm.fusion.MeshAggregator(P, C)
    twice.fuse_view(r, cams[10], probs[0])
    twice.fuse_view(r, cams[10], probs[0])
    np.testing.assert_allclose(twice.get_raw(), 2.0 * parts[0].get_raw(), rtol=1e-6)      # linearity
    out = whole.get()
    touched = out.sum(axis=1) > 0.5
    np.testing.assert_allclose(out[touched].sum(axis=1), 1.0, rtol=1e-5)
    assert (out[~touched] == 0).all() and touched.sum() > 600_000
    # mass conservation: with iew = 0 every valid pixel contributes its whole probability vector
    plain = sm.fusion.MeshAggregator(P, C, "sum", 0.0)
    plain.fuse_view(r, cams[10], probs[0])
    hp = np.asarray(probs[0]).reshape(-1, C)
    keep = (idx_a.reshape(-1) != BG) & (hp.sum(axis=1) > 0.5)
    np.testing.assert_allclose(plain.get_raw().astype(np.float64).sum(), hp[keep].astype(np.float64).sum(), rtol=1e-5)


def test_annotation_renderer_matches_oracle(sm, oracle):
    mesh, cams = small_scene()
    C, P = 6, len(mesh.faces)
    rng = np.random.default_rng(9)
    r = sm.render.triangles(mesh)
    agg, oagg = sm.fusion.MeshAggregator(P, C), oracle.OracleAggregator(P, C)
    idx, _ = r.render(cams[0])
    probs = random_probs(rng, *cams[0].resolution, C)
    agg.add(idx, probs)
    oagg.add(np.asarray(idx), probs)
    bg = np.linspace(-1, 1, C).astype(np.float32)
    idx2, _ = r.render(cams[1])
    got = agg.renderer().render(idx2, bg)                                  # device indices
    want = oracle.render_annotations(oagg, np.asarray(idx2), bg)
    assert got.shape == cams[1].resolution + (C,) and got.dtype == np.float32
    assert_fused_close(got, want)
    got_i64 = agg.renderer().render(np.asarray(idx2).astype(np.int64).T.copy().T, bg)   # host, int64, strided
    np.testing.assert_array_equal(got_i64, got)
    assert (got[np.asarray(idx2) == BG] == bg).all()
    np.testing.assert_array_equal(np.asarray(agg.renderer().render_device(idx2, bg)), got)   # the image left in HBM


@pytest.mark.parametrize("C,shape", [(1, (37, 29)), (19, (333, 257)), (150, (64, 48)), (300, (97, 5)), (7, (1, 1))])
def test_annotation_renderer_class_counts_and_odd_sizes(sm, C, shape):
    """k_gather_annotations walks 256-pixel blocks with an incremental (pixel, class) counter: class counts below, at and above
    the block size, images that are not a multiple of it.  Exact against a numpy gather of get()."""
    W, H = shape
    P = 50
    rng = np.random.default_rng(C)
    agg = sm.fusion.MeshAggregator(P, C)
    img = rng.integers(0, P, (W, H)).astype(np.uint32)
    img[rng.random((W, H)) < 0.2] = BG
    probs = random_probs(rng, W, H, C, 0.0)
    agg.add(img, probs)
    ann = agg.get()
    bg = rng.random(C, dtype=np.float32)
    got = agg.renderer().render(img, bg)
    want = np.where((img != BG)[..., None], ann[np.minimum(img, P - 1)], bg[None, None, :])
    np.testing.assert_array_equal(got, want.astype(np.float32))


def test_dlpack_export_feeds_add(sm, oracle):
    """render() output exported as a DLPack capsule (as the reference returns it) is accepted by add()."""
    mesh, cams = small_scene()
    C, P = 5, len(mesh.faces)
    rng = np.random.default_rng(4)
    r = sm.render.triangles(mesh)
    idx, _ = r.render(cams[0])
    probs = random_probs(rng, *cams[0].resolution, C)
    a1, a2 = sm.fusion.MeshAggregator(P, C), sm.fusion.MeshAggregator(P, C)
    a1.add(idx, probs)
    a2.add(idx.__dlpack__(), probs)          # capsule over HBM, kDLROCM
    np.testing.assert_allclose(a1.get_raw(), a2.get_raw(), rtol=1e-5, atol=1e-7)   # float atomics: order differs run to run


@pytest.mark.parametrize("kind", ["sum", "summax"])
@pytest.mark.parametrize("C", [5, 19, 40, 1, 7, 13, 21, 32, 33, 41, 48, 49, 64, 127, 150, 258])
def test_fuse_view_triangle_order_is_bit_exact(sm, oracle, kind, C):
    """smesh_fuse_view on a triangle renderer takes the triangle-order path (k_fuse_tri for C <= 40 -- exact instances for
    5 / 19 / 40, run-time-C instances otherwise --, k_fuse_tri_any up to 127, k_fuse_tri_wide beyond): every accumulator row has one owner and the reference's
    float32 operation order is kept, so the raw accumulator equals the float32 single-threaded oracle bit for bit
    (small triangles only; large ones are tree-reduced, see next test)."""
    mesh, cams = small_scene(120, 60, 320, 240, views=3)     # ~1.5 px triangles: all bounding boxes <= 8 x 8
    P = len(mesh.faces)
    rng = np.random.default_rng(C)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C, kind, 0.5)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oagg = oracle.OracleAggregator(P, C, kind, 0.5)             # float32 accumulators, 1 thread (conftest)
    for cam in cams:
        probs = random_probs(rng, *cam.resolution, C)
        weights = rng.random(cam.resolution, dtype=np.float32)
        agg.fuse_view(r, cam, probs, weights)
        oagg.add(o.render(cam)[0], probs, weights)
    np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))
    np.testing.assert_array_equal(agg.get().view(np.uint32), oagg.get().view(np.uint32))


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
def test_fuse_view_mixed_triangle_sizes(sm, oracle, kind):
    """Large triangles (bounding box > 8 x 8: cooperative k_fuse_big) next to small ones, device-resident probs."""
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(12, 6, 400, 300, views=3)        # ~40 px triangles
    extra_v = np.array([[-6, -4, -0.5], [6, -4, -0.5], [0, 5, -0.5]], np.float32)   # one huge triangle behind the grid
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    big = sm.data.Mesh(verts, faces)
    P, C = len(faces), 19
    rng = np.random.default_rng(1)
    r = sm.render.triangles(big)
    agg = sm.fusion.MeshAggregator(P, C, kind)
    o = oracle.OracleRenderer(verts, faces)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        for cam in cams:
            probs = random_probs(rng, *cam.resolution, C)
            if kind == "mul":
                probs = np.maximum(probs, 1e-3).astype(np.float32)
            agg.fuse_view(r, cam, to_device(probs))
            oidx = o.render(cam)[0]
            oagg.add(oidx, probs)
        assert (oidx == P - 1).sum() > 2000                     # the huge triangle is visible around the grid
        # Mul: bit-identical float32 terms on both sides, summed in double by k_fuse_tri / fuse_box ((hi, lo) state);
        # the hi-plane-only kernels behind SMESH_FUSE=strip sum thousands of pixels per primitive in float32
        import os
        mul_tol = 1e-5
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5 if kind != "mul" else mul_tol)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C", [3, 70, 130])
def test_fuse_view_any_class_count_mixed_triangle_sizes(sm, oracle, kind, C):
    """k_fuse_tri_any incl. its big-triangle waves (single-chunk and multi-chunk rows), all three aggregators."""
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(12, 6, 400, 300, views=2)
    extra_v = np.array([[-6, -4, -0.5], [6, -4, -0.5], [0, 5, -0.5]], np.float32)
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    P = len(faces)
    rng = np.random.default_rng(C)
    r = sm.render.triangles(sm.data.Mesh(verts, faces))
    agg = sm.fusion.MeshAggregator(P, C, kind)
    o = oracle.OracleRenderer(verts, faces)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        for cam in cams:
            probs = random_probs(rng, *cam.resolution, C)
            if kind == "mul":
                probs = np.maximum(probs, 1e-3).astype(np.float32)
            agg.fuse_view(r, cam, to_device(probs))
            oagg.add(o.render(cam)[0], probs)
        import os
        if os.environ.get("SMESH_FUSE") != "strip":
            assert sm._lib.last_fuse_kernel() == (
                "k_fuse_tri_wide" if C >= 128 and os.environ.get("SMESH_FUSE_WIDE") != "0" else "k_fuse_tri_any" if C > 48 else "k_fuse_tri")
        mul_tol = 1e-5   # (hi, lo) state in every triangle-order kernel; the generic scatter-add adds in float32 on the hi plane
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5 if kind != "mul" else mul_tol)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C,iew", [(5, 0.5), (19, 0.0), (19, 1.0), (33, 0.5), (48, 0.5)])
def test_fuse_views_medium_triangles(sm, oracle, kind, C, iew):
    """Meshes of MEDIUM triangles (boxes over 8 x 8, up to ~30 x 30 pixels: fuse_mid_entries in the k_fuse_tri launch, sixteen lanes per (triangle, view), float
    atomics; Mul: the tail waves of k_fuse_tri) through fuse_views with per-pixel weights, plus one triangle too large for sixteen
    lanes (tail wave) whose other views are medium, against the float64 oracle; the views of a group in one call and one by one."""
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(14, 9, 420, 310, views=5)          # ~20 x 20-pixel triangles
    extra_v = np.array([[-1.2, -0.9, 0.8], [1.2, -0.9, 0.8], [0, 1.0, 0.8]], np.float32)     # ~ 100 x 80 pixels: large in every view
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    P = len(faces)
    rng = np.random.default_rng(C * 7 + len(kind))
    r = sm.render.triangles(sm.data.Mesh(verts, faces))
    o = oracle.OracleRenderer(verts, faces)
    group, single = sm.fusion.MeshAggregator(P, C, kind, iew), sm.fusion.MeshAggregator(P, C, kind, iew)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind, iew)
        probs, weights = [], []
        for cam in cams:
            p = random_probs(rng, *cam.resolution, C)
            if kind == "mul":
                p = np.maximum(p, 1e-3).astype(np.float32)
            w = (rng.random(cam.resolution, dtype=np.float32) + 0.25).astype(np.float32)
            probs.append(p)
            weights.append(w)
            oagg.add(o.render(cam)[0], p, w)
        dp, dw = [to_device(p) for p in probs], [to_device(w) for w in weights]
        group.fuse_views(r, cams, dp, dw)
        for cam, p, w in zip(cams, dp, dw):
            single.fuse_view(r, cam, p, w)
        want = oagg.get()
        assert (want.sum(axis=1) > 0.5).sum() > P // 2
        # (Mul behind SMESH_FUSE=strip: the generic scatter-add adds in float32 on the hi plane, as in the tests above)
        import os
        tol = 1e-5
        assert_fused_close(group.get(), want, rtol=tol)
        assert_fused_close(single.get(), want, rtol=tol)
    finally:
        oracle.set_accum_double(False)


def test_fuse_view_user_index_images_take_generic_path(sm, oracle):
    mesh, cams = small_scene()
    P, C = len(mesh.faces) + 5, 7                               # aggregator larger than the mesh: not triangle-order
    rng = np.random.default_rng(3)
    r = sm.render.triangles(mesh)
    agg, oagg = sm.fusion.MeshAggregator(P, C), oracle.OracleAggregator(P, C)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    for cam in cams:
        probs = random_probs(rng, *cam.resolution, C)
        agg.fuse_view(r, cam, probs)
        oagg.add(o.render(cam)[0], probs)
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)


@pytest.mark.parametrize("cap", [1, 7, 300])
def test_render_fragment_queue_overflow(sm, oracle, monkeypatch, cap):
    """Tiny per-tile fragment queues: most fragments overflow into the global key image and the flagged tiles
    merge it; small and big triangles mixed, several renders in a row (queues, flags and keys re-armed)."""
    monkeypatch.setenv("SMESH_FRAG_CAP", str(cap))
    mesh, cams = small_scene(60, 30, 330, 250, views=3)
    extra_v = np.array([[-6, -4, -0.5], [6, -4, -0.5], [0, 5, -0.5]], np.float32)
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    r = sm.render.triangles(sm.data.Mesh(verts, faces))
    o = oracle.OracleRenderer(verts, faces)
    for cam in cams + cams[:1]:
        idx, depth = r.render(cam)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
    # the paired rasteriser launches of fuse_views (each view of a pair has its own queues, flags and key image)
    from semantic_meshes_amd.device import to_device
    P, C = len(faces), 19
    rng = np.random.default_rng(cap)
    agg = sm.fusion.MeshAggregator(P, C)
    probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
    agg.fuse_views(r, cams + cams[:1], [to_device(p) for p in probs + probs[:1]])
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for cam, p in zip(cams + cams[:1], probs + probs[:1]):
            oagg.add(o.render(cam)[0], p)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)
    monkeypatch.delenv("SMESH_FRAG_CAP")
    idx, depth = r.render(cams[1])                              # back to the default capacity (queues re-allocated)
    np.testing.assert_array_equal(np.asarray(idx), o.render(cams[1])[0])


@pytest.mark.parametrize("knob", ["SMESH_RASTER=direct", "SMESH_FUSE=strip", "SMESH_FUSE_WIDE=0",
                                  "SMESH_RASTER_PAIRS=0", "SMESH_FUSE_PAIRS=0", "SMESH_GROUP_PIPELINE=0", "SMESH_TEXEL_MULTI=0",
                                  "SMESH_RASTER_SPREAD=1", "SMESH_RASTER_WG_PUSH=1", "SMESH_RASTER_BALANCE=1 SMESH_RASTER_WG_PUSH=1"])
def test_alternative_paths_in_subprocess(knob):
    """These knobs are read once per process: re-run the render / fuse_view parity tests with the direct rasteriser
    (global 64-bit atomicMin per fragment), with the group pipeline off (the default since round 5 is on), and with the generic
    scatter-add forced behind fuse_view."""
    import os
    import subprocess
    import sys
    pairs = [kv.split("=") for kv in knob.split()]
    k, v = pairs[0]
    env = dict(os.environ, **{kk: vv for kk, vv in pairs})
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "render_small_scene or render_cfg1 or overlap or mixed_triangle or texel or fuse_view_cfg2"
    if k != "SMESH_FUSE":
        sel += " or triangle_order"
    if k in ("SMESH_RASTER", "SMESH_RASTER_PAIRS", "SMESH_FUSE_PAIRS", "SMESH_FUSE"):
        sel = "fuse_views" if k.endswith("PAIRS") else sel + " or fuse_views"
    if k == "SMESH_RASTER":
        sel += " or near_plane or room"      # the direct rasteriser clips at the near plane too
    if k in ("SMESH_RASTER_SPREAD", "SMESH_RASTER_WG_PUSH", "SMESH_RASTER_BALANCE"):     # (BALANCE: every cooperatively walked triangle through k_raster_medium)
        # the rasteriser's modes for views of medium triangles, forced on every view (spread waves: a triangle on nine lanes; the
        # view's queues filled by one atomic per workgroup): every kind of triangle through them, single views and groups
        sel += " or near_plane or medium or fuse_views or degenerate"
    if k in ("SMESH_GROUP_PIPELINE", "SMESH_TEXEL_MULTI"):
        sel = "fuse_views"
    # the many-instance multi-view tests only where the knob reaches them (each subprocess pays ~10 s of start-up as it is)
    drop = []
    if k not in ("SMESH_FUSE_WIDE", "SMESH_GROUP_PIPELINE", "SMESH_RASTER_PAIRS"):
        drop.append("wide_rows_equal")
    if k not in ("SMESH_TEXEL_MULTI", "SMESH_GROUP_PIPELINE", "SMESH_RASTER_PAIRS"):
        drop.append("texels_equal")
    if drop:
        sel = "(%s) and not (%s)" % (sel, " or ".join(drop))
    sel = "(%s) and not subprocess" % sel      # (-k is case-insensitive: "texel" would select the SMESH_TEXEL_MULTI instance of this very test)
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_parity.py"), "-q", "-x", "-m", "gpu",
                          "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


@pytest.mark.parametrize("C", [19, 64, 150])
def test_fuse_view_dont_care_threshold_is_exact(sm, oracle, C):
    """Row sums within a few ulps of the 0.5 don't-care threshold (Mesh.h:98): which pixels count is decided by the
    float32 sum in class order, whatever the kernel's own reduction order is (k_fuse_tri_wide replays the sequential
    sum when its tree estimate is too close to call; k_fuse_tri_any hands the running sum through the lane group)."""
    mesh, cams = small_scene(120, 60, 320, 240, views=2)
    P = len(mesh.faces)
    rng = np.random.default_rng(C + 100)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C, "sum", 0.5)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oagg = oracle.OracleAggregator(P, C, "sum", 0.5)
    flips = 0
    for cam in cams:
        W, H = cam.resolution
        probs = rng.random((W, H, C), dtype=np.float32) + 0.01
        seq = np.zeros((W, H), np.float32)
        for c in range(C):                                       # float32 sum in class order
            seq = seq + probs[:, :, c]
        probs *= (np.float32(0.5) / seq)[:, :, None]             # rows now sum to 0.5 up to rounding, on either side
        seq = np.zeros((W, H), np.float32)
        for c in range(C):
            seq = seq + probs[:, :, c]
        pair = probs.astype(np.float64).sum(-1)
        flips += int(((seq > 0.5) != (pair > 0.5)).sum())
        agg.fuse_view(r, cam, probs)
        oagg.add(o.render(cam)[0], probs)
    assert flips > 50                                            # the inputs do separate summation orders
    np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C", [4, 40])
def test_fuse_view_texels_small_triangles(sm, oracle, kind, C):
    """render.texels + fuse_view: triangle-order fusion over texel rows (k_fuse_texel); small triangles only, so the
    reference's float32 order is kept per texel row -> bit-equal to the single-threaded float32 oracle (Sum/Summax)."""
    import os
    mesh, cams = small_scene(160, 80, 330, 250, views=3)         # ~2 px triangles (all boxes <= 8 x 8), a few texels each
    r = sm.render.texels(mesh, cams, 1.5)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces, cams, 1.5)
    P = r.getPrimitivesNum()
    assert P > 2 * len(mesh.faces)
    rng = np.random.default_rng(C)
    agg, oagg = sm.fusion.MeshAggregator(P, C, kind, 0.5), oracle.OracleAggregator(P, C, kind, 0.5)
    for cam in cams:
        probs = random_probs(rng, *cam.resolution, C)
        if kind == "mul":
            probs = np.maximum(probs, 1e-3).astype(np.float32)
        weights = rng.random(cam.resolution, dtype=np.float32)
        agg.fuse_view(r, cam, probs, weights)
        oagg.add(o.render(cam)[0], probs, weights)
    if os.environ.get("SMESH_FUSE") != "strip":
        assert sm._lib.last_fuse_kernel() == "k_fuse_texel"
        if kind != "mul":
            np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)


@pytest.mark.parametrize("kind", ["sum", "summax"])
def test_fuse_view_texels_big_triangles(sm, oracle, kind):
    """Texel primitives on triangles larger than 8 x 8 pixels (cooperative waves: scratch histogram + float atomics
    confined to the triangle's own texel rows), mixed with small ones; several views in a row."""
    mesh, cams = small_scene(10, 5, 400, 300, views=3)            # ~50 px triangles
    fine, _ = small_scene(40, 20, 400, 300, views=1)
    verts = np.concatenate([mesh.vertices, fine.vertices + np.array([0, 0, 0.7], np.float32)])
    faces = np.concatenate([mesh.faces, fine.faces + len(mesh.vertices)]).astype(np.int32)
    both = sm.data.Mesh(verts, faces)
    r = sm.render.texels(both, cams, 0.3)
    o = oracle.OracleRenderer(verts, faces, cams, 0.3)
    P, C = r.getPrimitivesNum(), 7
    rng = np.random.default_rng(11)
    agg = sm.fusion.MeshAggregator(P, C, kind)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        for cam in cams:
            probs = random_probs(rng, *cam.resolution, C)
            agg.fuse_view(r, cam, probs)
            oagg.add(o.render(cam)[0], probs)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)
    # the scratch histogram is all zero again: a plain add() (generic path) still gives the right counts
    agg2, oagg2 = sm.fusion.MeshAggregator(P, C, kind), oracle.OracleAggregator(P, C, kind)
    probs = random_probs(rng, *cams[0].resolution, C)
    agg.reset()
    idx = r.render(cams[0])[0]
    agg.add(idx, probs)
    oagg2.add(np.asarray(idx), probs)
    assert_fused_close(agg.get(), oagg2.get(), rtol=1e-5)


@pytest.mark.parametrize("content_match", [False, True])
def test_add_after_render_takes_triangle_order_path(sm, oracle, monkeypatch, content_match):
    """The reference's two-call loop `idx, depth = renderer.render(cam); aggregator.add(idx, probs)`
    (colorize_cityscapes_mesh.py:65-67): add() recognises the untouched DeviceArray of one of the last six renders by identity and
    runs the triangle-order fusion on the records that render left ("render-records").  An image that went through another framework
    or numpy, an older render, an image that was changed: its records are rebuilt from the image ("image-records",
    image_records.hip) -- or, with SMESH_ADD_RECORDS_MIN_C above the class count (round 2's default below 32 classes;
    `content_match`), recognised by CONTENT ("render-records") and otherwise handed to the scatter-add.  All give the oracle's result."""
    import os
    if content_match:
        monkeypatch.setenv("SMESH_ADD_RECORDS_MIN_C", "32")
    else:
        monkeypatch.delenv("SMESH_ADD_RECORDS_MIN_C", raising=False)
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(120, 60, 320, 240, views=8)
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(5)
    r = sm.render.triangles(mesh)
    last = lambda: sm._lib.last_fuse_kernel()
    path = lambda: sm._lib.last_add_path()
    agg = sm.fusion.MeshAggregator(P, C, "sum", 0.5)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oagg = oracle.OracleAggregator(P, C, "sum", 0.5)
    forced_generic = os.environ.get("SMESH_FUSE") == "strip"
    fast = "k_scatter_strip" if forced_generic else "k_fuse_tri"
    matched = "scatter" if forced_generic else "render-records"                    # by identity
    records_off = forced_generic or os.environ.get("SMESH_ADD_RECORDS") == "0"       # (test_gpu_image_records.py runs this file that way too)
    content_match = content_match or records_off
    generic = "scatter" if content_match else "image-records"                      # no render recognised
    matched_c = matched if content_match else generic                              # a copy of a render: by content, or as any image
    for cam in cams[:3]:
        probs = random_probs(rng, *cam.resolution, C)
        weights = rng.random(cam.resolution, dtype=np.float32)
        idx, depth = r.render(cam)
        agg.add(idx, probs, weights)
        assert last() == fast and path() == matched
        oagg.add(o.render(cam)[0], probs, weights)
    if not forced_generic:
        np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))
    # 1. the last six renders are still recognised (the harness queues up to three views between its render and its add
    #    thread, eval_scannet.py:189-238); a seventh-oldest one is not: its per-triangle records are gone -> generic path
    probs = random_probs(rng, *cams[0].resolution, C)
    kept = [r.render(cams[k], lazy=False)[0] for k in range(7)]     # (rasterised NOW, in this order: a lazy plane would be rendered when add() looks at it)
    agg.add(kept[0], probs)
    assert path() == generic
    oagg.add(o.render(cams[0])[0], probs)
    for k in (1, 6, 3):
        agg.add(kept[k], probs)
        assert last() == fast and path() == matched
        oagg.add(o.render(cams[k])[0], probs)
    # 2. exported through __cuda_array_interface__: identity no longer proves anything, the content does
    _ = kept[5].__cuda_array_interface__
    assert kept[5]._exported
    agg.add(kept[5], probs)
    assert last() == fast and path() == matched_c
    oagg.add(o.render(cams[5])[0], probs)
    # ... and after someone changed a single pixel of it the generic path takes over (and honours the change)
    _ = kept[4].__cuda_array_interface__          # exported (the library takes the plane's checksum at this point) ...
    changed = np.asarray(kept[4]).copy()
    x, y = np.argwhere(changed != BG)[100]
    changed[x, y] = (changed[x, y] + 17) % P
    sm._lib.check(sm._lib.lib().smesh_memcpy(kept[4].ptr, changed.ctypes.data, changed.nbytes, sm._lib.MEM_DEVICE, sm._lib.MEM_HOST, 0))
    agg.add(kept[4], probs)                       # ... then modified in place by its new co-owner
    assert path() == generic
    oagg.add(changed, probs)
    # 3. a numpy COPY of a render (DLPack -> framework -> .numpy() in the reference's harness) with host probs
    agg.add(np.asarray(kept[2]), probs)
    assert last() == fast and path() == matched_c
    oagg.add(o.render(cams[2])[0], probs)
    agg.add(np.asarray(kept[2]).astype(np.int32), probs)          # int32 copies too (-1 == 0xFFFFFFFF)
    assert last() == fast and path() == matched_c
    oagg.add(o.render(cams[2])[0], probs)
    agg.add(np.asarray(kept[2]).astype(np.int64), probs)          # 64-bit images are never a copy of a plane: records from the image
    assert path() == generic
    oagg.add(o.render(cams[2])[0], probs)
    del kept
    # 4. device-resident probs and the latest render -> fast path again
    idx1b, _ = r.render(cams[1])
    agg.add(idx1b, to_device(probs))
    assert last() == fast and path() == matched
    oagg.add(o.render(cams[1])[0], probs)
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)


def test_harness_shaped_loop_takes_triangle_order_path(sm, oracle):
    """eval-scannet/eval_scannet.py:203-238 in miniature: the main thread renders, hands the planes to another framework
    through DLPack (torch stands in for TensorFlow), transposes them for display, and queues numpy copies (transposed back)
    three views deep; a worker adds them.  The index images reach add() as plain numpy arrays -- and are recognised by
    content, so the fusion runs in triangle order."""
    import os
    import queue
    import threading
    torch = pytest.importorskip("torch")
    mesh, cams = small_scene(120, 60, 320, 240, views=9)
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(15)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(primitives=P, classes=C, aggregator="sum", images_equal_weight=0.5)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oagg = oracle.OracleAggregator(P, C, "sum", 0.5)
    q = queue.Queue(maxsize=3)
    kernels, errors = [], []

    def worker():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                agg.add(*item)
                kernels.append(sm._lib.last_fuse_kernel())
        except Exception as e:   # surfaced by the main thread
            errors.append(e)

    t = threading.Thread(target=worker)
    t.start()
    for cam in cams:
        primitive_indices, depth = r.render(cam)
        hw = torch.from_dlpack(primitive_indices).transpose(0, 1)          # (H,W) for display, as the harness does
        assert hw.shape == (cam.resolution[1], cam.resolution[0])
        pred = random_probs(rng, cam.resolution[1], cam.resolution[0], C)  # network output, (H,W,C)
        q.put((hw.transpose(0, 1).contiguous().cpu().numpy(), np.ascontiguousarray(pred.transpose(1, 0, 2))))
        oagg.add(o.render(cam)[0], np.ascontiguousarray(pred.transpose(1, 0, 2)))
    q.put(None)
    t.join(120)
    assert not errors, errors
    assert len(kernels) == len(cams)
    if os.environ.get("SMESH_FUSE") != "strip":
        assert kernels == ["k_fuse_tri"] * len(cams), kernels
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)


@pytest.mark.parametrize("C", [5, 19, 40, 7, 150])
def test_fuse_view_small_and_big_triangles_interleaved(sm, oracle, C):
    """Triangles of 8-12 pixels: bounding boxes on both sides of the 8 x 8 limit inside the same 64-row accumulator
    block (regression: k_fuse_tri wrote the whole block back and overwrote the rows the big-triangle waves of the
    same launch were updating)."""
    mesh, cams = small_scene(100, 50, 640, 480, views=2)
    P = len(mesh.faces)
    rng = np.random.default_rng(C)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg = sm.fusion.MeshAggregator(P, C)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for cam in cams:
            probs = random_probs(rng, *cam.resolution, C)
            agg.fuse_view(r, cam, probs)
            oagg.add(o.render(cam)[0], probs)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("C", [19, 150])
def test_shuffled_face_order_is_reordered_internally(sm, oracle, C):
    """A mesh whose faces come in random order is processed in Morton order inside the library; the index image and the
    accumulator rows keep the caller's numbering.  Everything must equal the oracle run on the same (shuffled) mesh."""
    import os
    base, cams = small_scene(120, 60, 320, 240, views=3)
    rng = np.random.default_rng(42)
    perm = rng.permutation(len(base.faces))
    faces = np.ascontiguousarray(base.faces[perm])
    mesh = sm.data.Mesh(base.vertices, faces)
    P = len(faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(base.vertices, faces)
    agg, oagg = sm.fusion.MeshAggregator(P, C, "sum", 0.5), oracle.OracleAggregator(P, C, "sum", 0.5)
    for cam in cams:
        idx, depth = r.render(cam)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
        probs = random_probs(rng, *cam.resolution, C)
        agg.fuse_view(r, cam, probs)
        oagg.add(oidx, probs)
    if os.environ.get("SMESH_FUSE") != "strip":
        assert sm._lib.last_fuse_kernel() == (
            "k_fuse_tri" if C == 19 else "k_fuse_tri_wide" if os.environ.get("SMESH_FUSE_WIDE") != "0" else "k_fuse_tri_any")
        np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    # render() + add() on the re-ordered renderer, and bigger triangles (cooperative paths) with the id table
    coarse, cams2 = small_scene(60, 30, 640, 480, views=2)
    perm2 = rng.permutation(len(coarse.faces))
    faces2 = np.ascontiguousarray(coarse.faces[perm2])
    r2 = sm.render.triangles(sm.data.Mesh(coarse.vertices, faces2))
    o2 = oracle.OracleRenderer(coarse.vertices, faces2)
    agg2 = sm.fusion.MeshAggregator(len(faces2), C)
    oracle.set_accum_double(True)
    try:
        oagg2 = oracle.OracleAggregator(len(faces2), C)
        for cam in cams2:
            probs = random_probs(rng, *cam.resolution, C)
            idx, _ = r2.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), o2.render(cam)[0])
            agg2.add(idx, probs)
            oagg2.add(np.asarray(idx), probs)
        assert_fused_close(agg2.get(), oagg2.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_render_and_add_from_two_threads(sm, oracle):
    """The reference's harness renders on the main thread while a worker thread adds the previous view
    (python/scripts/eval_scannet.py:189-238); the GIL is released inside every entry point.  Same result as the
    sequential oracle, no deadlock."""
    import queue
    import threading
    mesh, cams = small_scene(120, 60, 320, 240, views=3)
    cams = cams * 4
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(9)
    all_probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    q = queue.Queue(maxsize=2)
    errors = []

    def worker():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                idx, probs = item
                agg.add(idx, probs)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=worker)
    t.start()
    for cam, probs in zip(cams, all_probs):
        idx, depth = r.render(cam)
        q.put((idx, probs))
    q.put(None)
    t.join(timeout=120)
    assert not t.is_alive() and not errors
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for cam, probs in zip(cams, all_probs):
            oagg.add(o.render(cam)[0], probs)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_degenerate_geometry(sm, oracle):
    """Zero-area, duplicate, out-of-range-index, behind-the-camera and non-finite triangles next to valid ones."""
    from semantic_meshes_amd import synth
    mesh, cams = small_scene(30, 15, 200, 150, views=2)
    v = mesh.vertices.copy()
    extra_v = np.array([[0, 0, 0], [0, 0, 0], [1, 1, 1], [np.nan, 0, 0], [1e30, 1e30, 1e30], [0.1, 0.2, 50.0]], np.float32)
    nv = len(v)
    verts = np.concatenate([v, extra_v])
    bad = np.array([[nv, nv + 1, nv + 2],          # two identical vertices: zero area
                    [0, 0, 0],                     # one vertex three times
                    [nv + 3, 1, 2],                # NaN coordinate
                    [nv + 4, 1, 2],                # absurd coordinate
                    [-1, 1, 2],                    # negative index
                    [len(verts) + 5, 1, 2],        # index past the end
                    [nv + 5, 3, 4]], np.int32)     # far behind / in front of nothing special
    faces = np.concatenate([mesh.faces[:200], bad, mesh.faces[200:], mesh.faces[:50]]).astype(np.int32)   # + exact duplicates
    with pytest.raises(ValueError):
        sm.data.Mesh(verts, faces)                 # the host-side mesh type refuses out-of-range indices ...
    import types
    r = sm.render.triangles(types.SimpleNamespace(vertices=verts, faces=faces))   # ... the library itself just skips such faces
    o = oracle.OracleRenderer(verts, faces)
    P, C = len(faces), 5
    rng = np.random.default_rng(3)
    agg, oagg = sm.fusion.MeshAggregator(P, C), oracle.OracleAggregator(P, C)
    for cam in cams:
        idx, depth = r.render(cam)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oidx)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
        probs = random_probs(rng, *cam.resolution, C)
        agg.fuse_view(r, cam, probs)
        oagg.add(oidx, probs)
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    # duplicates: the lower id wins everywhere (B-4), so the re-appended copies of faces 0..49 never show up
    assert not np.isin(np.asarray(idx), np.arange(P - 50, P)).any()


@pytest.mark.parametrize("res", [(1, 1), (7, 3), (33, 65), (64, 64), (65, 129), (1000, 9), (9, 1000)])
def test_odd_image_sizes(sm, oracle, res):
    """Images smaller than, equal to and not a multiple of the 32 x 64 screen tiles; extreme aspect ratios."""
    from semantic_meshes_amd import synth
    mesh = synth.grid_mesh(20, 10)
    W, H = res
    cams = [synth.ring_camera(k, 3, W, H) for k in range(3)]
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    P, C = len(mesh.faces), 5
    rng = np.random.default_rng(W * 1000 + H)
    agg = sm.fusion.MeshAggregator(P, C)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for cam in cams:
            idx, depth = r.render(cam)
            oidx, odepth = o.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oidx)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
            probs = random_probs(rng, W, H, C)
            agg.fuse_view(r, cam, probs)
            oagg.add(oidx, probs)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_get_device_matches_get(sm):
    mesh, cams = small_scene()
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(8)
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C, "mul")
    for cam in cams:
        agg.fuse_view(r, cam, np.maximum(random_probs(rng, *cam.resolution, C), 1e-3).astype(np.float32))
    host = agg.get()
    dev = agg.get_device()
    assert dev.shape == (P, C) and dev.dtype == np.float32
    np.testing.assert_array_equal(np.asarray(dev).view(np.uint32), host.view(np.uint32))


@pytest.mark.parametrize("P,C", [(1178401 // 7, 7), (308411, 19), (1048577, 2), (262147, 5), (4000037, 3)])
def test_get_into_pageable_memory_copies_every_byte(sm, P, C):
    """get() into PAGEABLE host memory of 4 MB and more goes through a ring of page-locked chunks and a few host threads that move each
    chunk on (fusion.hip copy_to_host / parallel_copy).  Round 4's sweep found the last n % 6 bytes of a chunk uncopied for sizes whose
    sixth is a multiple of 64 (4 713 604 bytes: one float of the result stale) -- here: awkward sizes, a known raw state, every element
    compared with the device-resident result."""
    import ctypes
    from semantic_meshes_amd import _lib
    rng = np.random.default_rng(P + C)
    agg = sm.fusion.MeshAggregator(P, C, "sum")
    raw = rng.random((P, C), dtype=np.float32) + 0.25
    agg.set_raw(raw)
    want = np.asarray(agg.get_device())
    out = np.full((P, C), np.float32(-7.0))          # pageable, pre-touched: a byte that is not copied shows
    _lib.check(_lib.lib().smesh_aggregator_get(agg._h, out.ctypes.data_as(ctypes.c_void_p), _lib.MEM_HOST))
    np.testing.assert_array_equal(out.view(np.uint32), want.view(np.uint32))
    np.testing.assert_allclose(out, raw / raw.sum(1, keepdims=True), rtol=1e-6)


def test_c99_client_against_the_hip_library(tmp_path):
    """tests/abi_smoke.c built with gcc against libsmesh_hip.so: the C ABI without any Python in between."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "semantic_meshes_amd", "csrc")
    exe = str(tmp_path / "abi_smoke_hip")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "abi_smoke.c"),
                           "-L", libdir, "-lsmesh_hip", "-lm", "-Wl,-rpath," + libdir, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi smoke ok" in out.stdout and "hip-gfx950" in out.stdout


def test_mul_aggregator_edge_values(sm, oracle):
    """Mul = sum of log(p^w) = w * log(p) (Fusion.cu:83-87): zero probabilities (-inf wipes the class out),
    zero weights (p^0 = 1 contributes nothing, also for p = 0), p = 1, and tiny probabilities with large weights."""
    mesh, cams = small_scene(60, 30, 200, 150, views=2)
    P, C = len(mesh.faces), 5
    rng = np.random.default_rng(77)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg = sm.fusion.MeshAggregator(P, C, "mul", 0.0)      # images_equal_weight 0: the pixel weight is the weights image
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, "mul", 0.0)
        for cam in cams:
            W, H = cam.resolution
            probs = random_probs(rng, W, H, C, zero_fraction=0.0)
            probs[rng.random((W, H)) < 0.2, 0] = 0.0                     # p = 0 in one class: log(0^w) = -inf
            probs[rng.random((W, H)) < 0.1, 1] = 1.0                     # p = 1: contributes exactly 0
            probs[rng.random((W, H)) < 0.1, 2] = 1e-30                   # w * log(p) stays finite (no pow: nothing underflows)
            weights = rng.choice(np.array([0.0, 0.5, 1.0, 3.0], np.float32), size=(W, H))
            agg.fuse_view(r, cam, probs, weights)
            oagg.add(o.render(cam)[0], probs, weights)
        got, want = agg.get(), oagg.get()
        assert np.isfinite(got).all()                                    # NaN / Inf -> 0 in get() (Fusion.h:79-95)
        assert_fused_close(got, want, rtol=1e-5, atol=1e-6)
        # the same classes are wiped out by -inf (threshold: exp() of -87 .. -103 is a denormal on the host, zero on the device)
        assert ((want < 1e-30) == (got < 1e-30)).mean() > 0.9999
    finally:
        oracle.set_accum_double(False)


def test_mul_against_the_literal_pow_then_log_oracle(sm, oracle):
    """the HIP Mul aggregator against the oracle's INDEPENDENT literal mode (logf(powf(p, w)) with libm, float64 sums):
    1e-5 on get() for ordinary class vectors -- a yardstick that did not move with the kernels."""
    mesh, cams = small_scene(60, 30, 200, 150, views=3)
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(78)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg = sm.fusion.MeshAggregator(P, C, "mul")
    oracle.set_accum_double(True)
    oracle.set_mul_literal(True)
    try:
        oagg = oracle.OracleAggregator(P, C, "mul")
        for cam in cams:
            W, H = cam.resolution
            probs = random_probs(rng, W, H, C)
            weights = (rng.random((W, H), dtype=np.float32) + 0.5).astype(np.float32)
            agg.fuse_view(r, cam, probs, weights)
            oagg.add(o.render(cam)[0], probs, weights)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_mul_literal(False)
        oracle.set_accum_double(False)


# ---- smesh_fuse_views: a batch of views, consecutive views fused two per launch ----------------------------------
@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C", [5, 19, 7, 40, 48])
def test_fuse_views_pairs_equal_single_calls_bit_for_bit(sm, oracle, kind, C):
    """fuse_views == fuse_view per view, in order: with small triangles only the two-views-per-launch kernel makes the
    same float32 additions in the same order, so the raw accumulators agree bit for bit (and with the float32 oracle for
    Sum / Summax).  Five views: two pairs and a single; weights on every view."""
    import os
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(120, 60, 320, 240, views=3)      # ~1.5 px triangles: all bounding boxes <= 8 x 8
    cams = cams + cams[:2]
    P = len(mesh.faces)
    rng = np.random.default_rng(100 + C)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    batch, single = sm.fusion.MeshAggregator(P, C, kind, 0.5), sm.fusion.MeshAggregator(P, C, kind, 0.5)
    probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
    if kind == "mul":
        probs = [np.maximum(p, 1e-3).astype(np.float32) for p in probs]
    weights = [rng.random(cam.resolution, dtype=np.float32) for cam in cams]
    dp, dw = [to_device(p) for p in probs], [to_device(w) for w in weights]
    batch.fuse_views(r, cams, dp, dw)
    # Sum / Summax: the float32 oracle (bit-equality below).  Mul: the float64-accumulating oracle is the yardstick -- float32
    # sums of log-probabilities added pixel by pixel (the reference's LogProb<float>) are the less accurate side.
    oracle.set_accum_double(kind == "mul")
    try:
        oagg = oracle.OracleAggregator(P, C, kind, 0.5)
        for k, cam in enumerate(cams):
            single.fuse_view(r, cam, dp[k], dw[k])
            oagg.add(o.render(cam)[0], probs[k], weights[k])
        want = oagg.get()
        oraw = None if kind == "mul" else oagg.get_raw()
    finally:
        oracle.set_accum_double(False)
    if os.environ.get("SMESH_FUSE") != "strip":
        assert sm._lib.last_fuse_kernel() == "k_fuse_tri"
        np.testing.assert_array_equal(batch.get_raw().view(np.uint32), single.get_raw().view(np.uint32))
        if kind != "mul":
            np.testing.assert_array_equal(batch.get_raw().view(np.uint32), oraw.view(np.uint32))
    assert_fused_close(batch.get(), want, rtol=1e-5)


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C", [150, 300, 520, 64, 100])
def test_fuse_views_wide_rows_equal_single_calls_bit_for_bit(sm, oracle, kind, C):
    """Rows beyond k_fuse_tri -- k_fuse_tri_wide (one / two / four 16-byte pieces per lane) and k_fuse_tri_any (C = 64 / 100: a row over
    two / four lanes): up to eight views per launch, a triangle's row making one round trip for all of them -- the same float32
    additions in the same order as one call per view, so the raw accumulators agree bit for bit with each other and (Sum / Summax)
    with the float32 oracle.  Eleven views: launches of 8, 2 and 1."""
    import os
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(80, 40, 200, 150, views=4)        # ~1.5 px triangles: all bounding boxes <= 8 x 8
    cams = cams + cams[:3] + cams
    P = len(mesh.faces)
    rng = np.random.default_rng(200 + C)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    batch, single = sm.fusion.MeshAggregator(P, C, kind, 0.5), sm.fusion.MeshAggregator(P, C, kind, 0.5)
    probs = [random_probs(rng, *cam.resolution, C, 0.03) for cam in cams[:4]]
    if kind == "mul":
        probs = [np.maximum(p, 1e-3).astype(np.float32) for p in probs]
    probs = probs + probs[:3] + probs
    weights = [rng.random(cam.resolution, dtype=np.float32) for cam in cams]
    dp, dw = [to_device(p) for p in probs[:4]], [to_device(w) for w in weights]
    dp = dp + dp[:3] + dp
    batch.fuse_views(r, cams, dp, dw)
    oracle.set_accum_double(kind == "mul")
    try:
        oagg = oracle.OracleAggregator(P, C, kind, 0.5)
        oidx = [o.render(cam)[0] for cam in cams[:4]]
        oidx = oidx + oidx[:3] + oidx
        for k, cam in enumerate(cams):
            single.fuse_view(r, cam, dp[k], dw[k])
            oagg.add(oidx[k], probs[k], weights[k])
        want = oagg.get()
        oraw = None if kind == "mul" else oagg.get_raw()
    finally:
        oracle.set_accum_double(False)
    if os.environ.get("SMESH_FUSE") != "strip":
        wide = C >= 128 and os.environ.get("SMESH_FUSE_WIDE") != "0"
        assert sm._lib.last_fuse_kernel() == ("k_fuse_tri_wide" if wide else "k_fuse_tri_any")
        if kind != "mul":      # (Mul: the hi plane is re-centred once per launch, so the grouping shows in the last bits)
            np.testing.assert_array_equal(batch.get_raw().view(np.uint32), single.get_raw().view(np.uint32))
            np.testing.assert_array_equal(batch.get_raw().view(np.uint32), oraw.view(np.uint32))
    mul_tol = 1e-5   # Mul: (hi, lo) state, a view's terms summed in double (float64 atomics + one fold per row on the generic path), against the float64 oracle
    assert_fused_close(batch.get(), want, rtol=mul_tol if kind == "mul" else 1e-5)
    if kind == "mul":
        assert_fused_close(batch.get(), single.get(), rtol=mul_tol)


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
@pytest.mark.parametrize("C,tpp", [(4, 1.5), (19, 2.2), (40, 0.8)])
def test_fuse_views_texels_equal_single_calls_bit_for_bit(sm, oracle, kind, C, tpp):
    """Texel renderers through fuse_views: the views of a group in ONE launch of k_fuse_texel_multi, a triangle's texel rows kept in
    registers across its pixels and views -- per row the additions of one call per view, in their order: raw accumulators bit-equal
    to the per-view path and (Sum / Summax) to the float32 oracle.  Eleven views: groups of 8 and 3; texel resolutions 1 .. 3."""
    import os
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(160, 80, 330, 250, views=4)       # ~2 px triangles: every box <= 8 x 8 (no atomics anywhere), a few texels each
    r = sm.render.texels(mesh, cams, tpp)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces, cams, tpp)
    cams = cams + cams[:3] + cams
    P = r.getPrimitivesNum()
    assert P == o.getPrimitivesNum() and P > len(mesh.faces)
    rng = np.random.default_rng(300 + C)
    batch, single = sm.fusion.MeshAggregator(P, C, kind, 0.5), sm.fusion.MeshAggregator(P, C, kind, 0.5)
    probs = [random_probs(rng, *cam.resolution, C) for cam in cams[:4]]
    if kind == "mul":
        probs = [np.maximum(p, 1e-3).astype(np.float32) for p in probs]
    probs = probs + probs[:3] + probs
    weights = [rng.random(cam.resolution, dtype=np.float32) for cam in cams]
    dp, dw = [to_device(p) for p in probs[:4]], [to_device(w) for w in weights]
    dp = dp + dp[:3] + dp
    batch.fuse_views(r, cams, dp, dw)
    assert sm._lib.last_fuse_kernel() in ("k_fuse_texel", "k_scatter_strip", "k_scatter_flat")   # (SMESH_FUSE=strip: Mul takes the float64-atomic flat path)
    oracle.set_accum_double(kind == "mul")
    try:
        oagg = oracle.OracleAggregator(P, C, kind, 0.5)
        oidx = [o.render(cam)[0] for cam in cams[:4]]
        oidx = oidx + oidx[:3] + oidx
        for k, cam in enumerate(cams):
            single.fuse_view(r, cam, dp[k], dw[k])
            oagg.add(oidx[k], probs[k], weights[k])
        want = oagg.get()
        oraw = None if kind == "mul" else oagg.get_raw()
    finally:
        oracle.set_accum_double(False)
    if os.environ.get("SMESH_FUSE") != "strip" and kind != "mul":
        np.testing.assert_array_equal(batch.get_raw().view(np.uint32), single.get_raw().view(np.uint32))
        np.testing.assert_array_equal(batch.get_raw().view(np.uint32), oraw.view(np.uint32))
    mul_tol = 1e-5     # Mul: (hi, lo) rows, every term folded in double, against the float64 oracle
    assert_fused_close(batch.get(), want, rtol=mul_tol if kind == "mul" else 1e-5)
    assert_fused_close(batch.get(), single.get(), rtol=mul_tol if kind == "mul" else 1e-5)


def test_fuse_views_texels_with_big_triangles(sm, oracle):
    """Texel renderer, a group whose views see the same triangles small in one view and with a box over 8 x 8 in another: the small
    records go through k_fuse_texel_multi, the big ones through each view's k_fuse_texel_big behind it."""
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.device import to_device
    mesh = synth.grid_mesh(40, 20)
    cams = [synth.ring_camera(k, 5, w, h) for k, (w, h) in enumerate([(400, 300), (160, 120), (640, 480), (400, 300), (160, 120)])]
    r = sm.render.texels(mesh, cams, 0.3)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces, cams, 0.3)
    P, C = r.getPrimitivesNum(), 19
    rng = np.random.default_rng(23)
    for kind in ("sum", "summax"):
        agg = sm.fusion.MeshAggregator(P, C, kind)
        probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
        agg.fuse_views(r, cams, [to_device(p) for p in probs])
        oracle.set_accum_double(True)
        try:
            oagg = oracle.OracleAggregator(P, C, kind)
            for k, cam in enumerate(cams):
                oagg.add(o.render(cam)[0], probs[k])
            assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
        finally:
            oracle.set_accum_double(False)


@pytest.mark.parametrize("C", [150, 70])
def test_fuse_views_wide_rows_mixed_triangle_sizes(sm, oracle, C):
    """C = 150 (k_fuse_tri_wide) / 70 (k_fuse_tri_any) with triangles that are small in some views of a launch and big (box over 8 x 8) in others, one huge triangle, views of
    different resolutions, a re-ordered mesh: rows of triangles that are big anywhere go to one wave of k_fuse_big_any for all views."""
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.device import to_device
    mesh = synth.grid_mesh(60, 30)
    extra_v = np.array([[-6, -4, -0.5], [6, -4, -0.5], [0, 5, -0.5]], np.float32)
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    rng = np.random.default_rng(17)
    for shuffled in (False, True):
        if shuffled:
            faces = np.ascontiguousarray(faces[rng.permutation(len(faces))])
        P = len(faces)
        cams = [synth.ring_camera(k, 7, w, h) for k, (w, h) in enumerate([(400, 300), (200, 150), (640, 480), (400, 300), (200, 150),
                                                                            (400, 300), (320, 240)])]
        r = sm.render.triangles(sm.data.Mesh(verts, faces))
        o = oracle.OracleRenderer(verts, faces)
        for kind in ("sum", "summax"):
            agg = sm.fusion.MeshAggregator(P, C, kind)
            probs = [random_probs(rng, *cam.resolution, C, 0.03) for cam in cams]
            weights = [rng.random(cam.resolution, dtype=np.float32) for cam in cams]
            agg.fuse_views(r, cams, [to_device(p) for p in probs], [to_device(w) for w in weights])
            oracle.set_accum_double(True)
            try:
                oagg = oracle.OracleAggregator(P, C, kind)
                for k, cam in enumerate(cams):
                    oagg.add(o.render(cam)[0], probs[k], weights[k])
                assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
            finally:
                oracle.set_accum_double(False)


@pytest.mark.parametrize("kind", ["sum", "summax"])
def test_fuse_views_mixed_triangle_sizes_and_image_sizes(sm, oracle, kind):
    """Pairs in which a triangle is small in one view and big (bounding box over 8 x 8) in the other, plus one huge
    triangle: those rows are handed to one tail wave for both views.  The two views of a pair differ in resolution."""
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.device import to_device
    mesh = synth.grid_mesh(100, 50)
    extra_v = np.array([[-6, -4, -0.5], [6, -4, -0.5], [0, 5, -0.5]], np.float32)
    verts = np.concatenate([mesh.vertices, extra_v])
    faces = np.concatenate([mesh.faces, [[len(mesh.vertices), len(mesh.vertices) + 1, len(mesh.vertices) + 2]]]).astype(np.int32)
    big = sm.data.Mesh(verts, faces)
    P, C = len(faces), 19
    # 8-12 px triangles at 640x480 (both sides of the limit), ~3 px at 320x240, ~30 px at 1280x960
    cams = [synth.ring_camera(k, 6, w, h) for k, (w, h) in enumerate([(640, 480), (320, 240), (1280, 960), (640, 480),
                                                                        (320, 240), (640, 480)])]
    rng = np.random.default_rng(7)
    r = sm.render.triangles(big)
    o = oracle.OracleRenderer(verts, faces)
    agg = sm.fusion.MeshAggregator(P, C, kind)
    probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
    agg.fuse_views(r, cams, [to_device(p) for p in probs])
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        for k, cam in enumerate(cams):
            oagg.add(o.render(cam)[0], probs[k])
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_fuse_views_shuffled_faces_and_fallbacks(sm, oracle):
    """Re-ordered mesh (per-lane rows) in pairs; host images, class counts beyond k_fuse_tri and texel renderers take the
    one-view-at-a-time path behind the same call."""
    import os
    from semantic_meshes_amd.device import to_device
    base, cams = small_scene(120, 60, 320, 240, views=4)
    rng = np.random.default_rng(3)
    faces = np.ascontiguousarray(base.faces[rng.permutation(len(base.faces))])
    mesh = sm.data.Mesh(base.vertices, faces)
    P = len(faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(base.vertices, faces)
    oidx = [o.render(cam)[0] for cam in cams]
    for C, on_device in ((19, True), (19, False), (64, True), (150, True)):
        agg, oagg = sm.fusion.MeshAggregator(P, C, "sum", 0.5), oracle.OracleAggregator(P, C, "sum", 0.5)
        probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
        agg.fuse_views(r, cams, [to_device(p) for p in probs] if on_device else probs)
        for k in range(len(cams)):
            oagg.add(oidx[k], probs[k])
        if os.environ.get("SMESH_FUSE") != "strip":
            np.testing.assert_array_equal(agg.get_raw().view(np.uint32), oagg.get_raw().view(np.uint32))
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    # texel renderer
    tmesh, tcams = small_scene(60, 30, 330, 250, views=3)
    tr = sm.render.texels(tmesh, tcams, 0.6)
    to = oracle.OracleRenderer(tmesh.vertices, tmesh.faces, tcams, 0.6)
    TP = tr.getPrimitivesNum()
    agg, oagg = sm.fusion.MeshAggregator(TP, 4), oracle.OracleAggregator(TP, 4)
    probs = [random_probs(rng, *cam.resolution, 4) for cam in tcams]
    agg.fuse_views(tr, tcams, [to_device(p) for p in probs])
    for k, cam in enumerate(tcams):
        oagg.add(to.render(cam)[0], probs[k])
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    # argument checks
    with pytest.raises(ValueError):
        agg.fuse_views(tr, tcams, probs[:2])
    agg.fuse_views(tr, [], [])


def test_fuse_views_sharded_on_the_device(sm, oracle):
    """distributed.fuse_views_sharded with the HIP aggregator hands this rank's views over eight per call (fuse_views); without a
    process group it is the whole job: equal to fusing every view, host or device images alike."""
    from semantic_meshes_amd import distributed as smdist
    from semantic_meshes_amd.device import to_device
    mesh, cams = small_scene(120, 60, 320, 240, views=11)
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(11)
    probs = [random_probs(rng, *cam.resolution, C) for cam in cams]
    dprobs = [to_device(p) for p in probs]
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg, hagg = sm.fusion.MeshAggregator(P, C), sm.fusion.MeshAggregator(P, C)
    smdist.fuse_views_sharded(r, agg, cams, lambda k: dprobs[k], batch=4)
    smdist.fuse_views_sharded(r, hagg, cams, lambda k: probs[k])
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C)
        for k, cam in enumerate(cams):
            oagg.add(o.render(cam)[0], probs[k])
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
        assert_fused_close(hagg.get(), oagg.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_huge_stage_is_left_out_only_where_it_is_provably_empty(sm, oracle):
    """The launch for near-plane-crossing / over-64-pixel triangles is skipped when the library can prove its queue empty from the
    mesh's bounding box and longest edge (raster.hip no_huge_possible).  A camera approaching a coarse grid from far away, head on
    and obliquely, crosses that proof's boundary: wherever the launch is declared unnecessary the queue really is empty, the
    declaration is used on both sides of the boundary, and the render is bit-equal to the oracle throughout."""
    from semantic_meshes_amd import synth
    mesh = synth.grid_mesh(24, 18, extent=10.0, relief=0.6)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    skipped = needed = no_big = 0
    for k, (dist, elev, off) in enumerate([(d, e, s) for d in (400.0, 120.0, 60.0, 30.0, 14.0, 7.0, 3.0)
                                           for e in (80.0, 35.0, 8.0) for s in (0.0, 4.5)]):
        th = 0.7 * k
        el = np.radians(elev)
        eye = (off + dist * np.cos(th) * np.cos(el), dist * np.sin(th) * np.cos(el), dist * np.sin(el))
        R, t = synth.look_at(eye, (off, 0.0, 0.0))
        W, H = (320, 240) if k % 2 else (640, 400)
        cam = sm.data.Camera(R, t, np.asarray([W, H]), np.asarray([0.9 * W, 0.8 * W]), np.asarray([W / 2.0 - 7.0, H / 2.0 + 3.0]))
        idx, depth = r.render_numpy(cam)
        need, q = r.render_stats(cam)
        if need:
            needed += 1
        else:
            skipped += 1
            assert q[2] == 0, (dist, elev, off, q)
        if not r.last_big_stage_needed:                 # proven: no box over 8 x 8 pixels -> nothing queued at all
            no_big += 1
            assert q[0] == 0 and q[3] == 0 and not need, (dist, elev, off, q)
        oidx, odepth = o.render(cam)
        np.testing.assert_array_equal(idx, oidx)
        np.testing.assert_array_equal(depth.view(np.uint32), odepth.view(np.uint32))
    assert skipped >= 8 and needed >= 8 and no_big >= 4, (skipped, needed, no_big)
    # cfg2's cameras: provably no such triangle (the headline workload does not pay for the launch)
    mesh2, cams2, _ = synth.scene("cfg2")
    r2 = sm.render.triangles(mesh2)
    assert not any(r2.render_stats(c, queues=False)[0] for c in cams2[::17])


WIDE_LIST_SCRIPT = r"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SMESH_ROOT"]); sys.path.insert(0, os.path.join(os.environ["SMESH_ROOT"], "tests"))
import semantic_meshes_amd as sm
from semantic_meshes_amd import synth, _lib
from helpers import small_scene
out = {}
for name, (ga, gb) in (("fine", (170, 81)), ("six_pixel_boxes", (50, 25))):
    mesh, cams = small_scene(ga, gb, 320, 240, views=11)
    P = len(mesh.faces)
    r = sm.render.triangles(mesh)
    for C in (150, 131):
        probs = [synth.device_probs(320, 240, C, synth.probs_seed(9, k), 0.05, 0) for k in range(len(cams))]
        agg = sm.fusion.MeshAggregator(P, C)
        agg.fuse_views(r, cams[:8], probs[:8])
        agg.fuse_views(r, cams[8:], probs[8:])
        agg.defer = False
        agg.fuse_view(r, cams[0], probs[0])
        assert _lib.last_fuse_kernel() == "k_fuse_tri_wide"
        out["%s_%d" % (name, C)] = agg.get_raw()
np.savez(sys.argv[1], **out)
"""


def test_wide_rows_pixel_list_kernel_equals_the_stream_kernel_bit_for_bit(tmp_path, sm):
    """k_fuse_tri_wide_list (round 6: Sum, 128 <= C < 256 -- the wave's pixels written to a list in LDS first, then walked through a ring of loads)
    against k_fuse_tri_wide (SMESH_WIDE_LIST=0): per accumulator row the same float32 additions in the same order, so the raw accumulators are
    bit-equal -- on a fine mesh (a few entries per triangle) and on one of six-pixel boxes seen in eight views (thousands of entries per wave:
    the list is walked in several chunks), C = 150 (a row tail of two classes) and 131 (of three)."""
    import os
    import subprocess
    import sys
    script = os.path.join(tmp_path, "wide.py")
    open(script, "w").write(WIDE_LIST_SCRIPT)
    got = {}
    for knob in ("0", "1"):
        path = os.path.join(tmp_path, "raw%s.npz" % knob)
        env = dict(os.environ, SMESH_WIDE_LIST=knob, SMESH_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        res = subprocess.run([sys.executable, script, path], env=env, capture_output=True, text=True, timeout=600)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
        got[knob] = np.load(path)
    for key in got["0"].files:
        a, b = got["0"][key], got["1"][key]
        assert (a != 0).sum() > a.size // 8, key
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), key
