"""CPU tests of the oracle: analytic known answers derived from the reference's in-tree lines
 and the committed golden fixtures.  No GPU needed."""
import os

import numpy as np
import pytest

from helpers import BG

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _camera(W=64, H=48, eye=(0, 0, 5.0), target=(0, 0, 0), f=60.0):
    from semantic_meshes_amd import data, synth
    R, t = synth.look_at(eye, target, up=(0, 1, 0))
    return data.Camera(R, t, np.array([W, H]), np.array([f, f]), np.array([W / 2.0, H / 2.0]))


def _unrle(vals, lens, shape):
    return np.repeat(vals, lens).reshape(shape)


# ---- rasteriser -----------------------------------------------------------------------------------
def near_plane_scene():
    """Raster spec B-3: triangles that cross the near plane z_c = 1e-6 are CLIPPED against it.
    Camera at the origin looking down +z; triangle 0 lies in front, triangle 1 (nearer, covering triangle 0) has one vertex
    behind the camera, triangle 2 has a vertex exactly on the camera plane."""
    from semantic_meshes_amd import data
    cam = data.Camera(np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.array([64, 48]), np.array([40.0, 40.0]), np.array([32.0, 24.0]))
    v = np.array([[-1, -1, 4], [1, -1, 4], [0, 1, 4],          # 0: in front
                  [-2, -2, 2], [0.5, -2, 2], [-0.5, 2, -1],    # 1: crosses the camera plane
                  [0.6, 0.2, 3], [1.8, 1.4, 3], [1.6, 0.9, 0]], np.float32)   # 2: touches it (z_c == 0)
    f = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    return cam, v, f


def room_scene(W=96, H=72, eye=(0.3, -0.2, 0.1), target=(2.0, 0.5, 0.0), f=50.0, half=(2.0, 1.5, 1.2)):
    """The camera INSIDE a closed box of twelve triangles (cfg4's namesake: a ScanNet room seen from within,
    eval-scannet/eval_scannet.py:203-238): every wall but the one behind the camera crosses the camera plane."""
    from semantic_meshes_amd import data, synth
    hx, hy, hz = half
    v = np.array([[sx * hx, sy * hy, sz * hz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    faces = np.array([t for q in quads for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3]))], np.int32)
    R, t = synth.look_at(eye, target, up=(0, 0, 1))
    cam = data.Camera(R, t, np.array([W, H]), np.array([f, f]), np.array([W / 2.0, H / 2.0]))
    return cam, v, faces


def test_near_plane_clipping_known_answers(oracle):
    cam, v, f = near_plane_scene()
    idx, depth = oracle.OracleRenderer(v, f).render(cam)
    assert set(np.unique(idx)) == {0, 1, 2, BG}
    # what is left of triangle 1 in front of the camera hides triangle 0 (z = 4) wherever it covers it; its depth is its
    # plane's z_c along the pixel's ray, anywhere between the near plane and 2
    assert (idx == 1).sum() > 800 and depth[idx == 1].max() <= 2.0 + 1e-6 and depth[idx == 1].min() >= 1e-6
    assert 0 < (idx == 0).sum() < 200 and (idx == 2).sum() > 50
    # against the independent ray caster: same primitive at every pixel that is not within a hair of an edge, same depth
    from helpers import raycast
    want, want_z, _, _, margin = raycast(cam, v, f)
    sure = margin > 1e-9
    assert sure.mean() > 0.97
    assert np.array_equal(idx[sure], want[sure])
    hit = sure & (want != BG)
    np.testing.assert_allclose(depth[hit], want_z[hit], rtol=2e-6)
    # moved in front of the plane nothing is clipped any more: same picture from the unclipped path up to the moved geometry
    v2 = v.copy()
    v2[5, 2] = 0.5
    idx2, _ = oracle.OracleRenderer(v2, f).render(cam)
    assert (idx2 == 1).sum() > 500


def test_camera_inside_a_room_has_no_holes(oracle):
    from helpers import raycast
    for eye, target in (((0.3, -0.2, 0.1), (2.0, 0.5, 0.0)), ((-1.2, 0.9, -0.6), (0.0, -1.5, 0.4)), ((1.5, 1.2, 1.0), (-2.0, -1.5, -1.2))):
        cam, v, f = room_scene(eye=eye, target=target)
        idx, depth = oracle.OracleRenderer(v, f).render(cam)
        assert (idx != BG).all() and np.isfinite(depth).all()          # a closed room: no pixel sees the background
        want, want_z, _, _, margin = raycast(cam, v, f)
        sure = margin > 1e-9
        assert sure.mean() > 0.95
        assert np.array_equal(idx[sure], want[sure])
        np.testing.assert_allclose(depth[sure], want_z[sure], rtol=2e-6)
        # vertex order and orientation of the crossing triangles do not matter
        g = f[:, [1, 2, 0]].copy()
        g[::2] = g[::2, ::-1]
        idx_g, depth_g = oracle.OracleRenderer(v, g).render(cam)
        assert np.array_equal(idx_g[sure], idx[sure]) and (idx_g != BG).all()
        np.testing.assert_allclose(depth_g[sure], depth[sure], rtol=2e-6)


def test_clipped_texel_triangles_number_their_texels_by_true_barycentrics(oracle):
    """Texel primitives of triangles that cross the near plane: the texel under a pixel is the one its ray hits (perspective-correct
    barycentric coordinates of the ORIGINAL triangle), the same rule texel_index applies to unclipped triangles' weights."""
    from helpers import raycast
    cam, v, f = room_scene(W=128, H=96)
    ctor_cam = _camera(W=128, H=96, eye=(0.0, 0.0, 9.0), target=(0, 0, 0), f=40.0)    # a camera that sees the room from outside sizes the textures
    r = oracle.OracleRenderer(v, f, cameras=[ctor_cam], texels_per_pixel=0.35)
    faces_r, res, first = r.texel_layout()
    assert res.max() >= 4
    idx, depth = r.render(cam)
    want_tri, want_z, b1, b2, margin = raycast(cam, v, faces_r)
    # the triangles this is about: those with a vertex at or behind the near plane (unclipped ones keep the screen-space weights)
    vc = (np.asarray(cam.rotation, np.float64) @ v.T.astype(np.float64)).T + np.asarray(cam.translation, np.float64)
    crossing = (vc[faces_r][:, :, 2] <= 1e-6).any(axis=1)
    assert crossing.sum() >= 6
    sure = (margin > 1e-6) & (want_tri != BG) & crossing[np.minimum(want_tri, len(faces_r) - 1)]
    tri = want_tri[sure]
    rr = res[tri].astype(np.int64)
    tu = np.clip(((b1[sure].astype(np.float32) - np.float32(1e-6)) * rr.astype(np.float32)).astype(np.int64), 0, None)
    tv = np.clip(((b2[sure].astype(np.float32) - np.float32(1e-6)) * rr.astype(np.float32)).astype(np.int64), 0, None)
    tu = np.minimum(tu, rr - 1)
    tv = np.minimum(tv, rr - 1 - tu)
    row = tu + tv
    want = first[tri].astype(np.int64) + row * (row + 1) // 2 + tu
    got = idx[sure].astype(np.int64)
    has = rr > 0
    # the texel changes where (b - 1e-6) * r crosses an integer: allow the pixels within rounding of such a border
    fu = (b1[sure] - 1e-6) * rr
    fv = (b2[sure] - 1e-6) * rr
    border = (np.abs(fu - np.round(fu)) < 1e-4) | (np.abs(fv - np.round(fv)) < 1e-4)
    ok = has & ~border
    assert ok.sum() > 0.8 * sure.sum()
    assert np.array_equal(got[ok], want[ok])


def test_ka1_empty_scene_is_background(oracle):
    r = oracle.OracleRenderer(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32))
    idx, depth = r.render(_camera())
    assert idx.shape == (64, 48) and idx.dtype == np.uint32
    assert (idx == BG).all() and np.isposinf(depth).all()        # TriangleRenderer.h:75-78


def test_ka2_single_triangle_layout_and_depth(oracle):
    v = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0]], np.float32)
    r = oracle.OracleRenderer(v, np.array([[0, 1, 2]], np.int32))
    assert r.getPrimitivesNum() == 1
    idx, depth = r.render(_camera())
    cov = idx == 0
    assert 200 < cov.sum() < 400 and ((idx == 0) | (idx == BG)).all()
    np.testing.assert_allclose(depth[cov], 5.0, rtol=1e-6)       # camera-space z of the plane z=0 seen from z=5
    assert np.isposinf(depth[~cov]).all()
    # (W,H) layout, y fastest: the triangle is wider at the bottom... image y grows downwards (camera +y is down)
    cols = np.flatnonzero(cov.any(axis=1))
    rows = np.flatnonzero(cov.any(axis=0))
    assert cols.min() >= 18 and cols.max() <= 45 and rows.min() >= 10 and rows.max() <= 37


def test_ka3_nearest_wins_and_vertex_order_is_irrelevant(oracle):
    v = np.array([[-1, -1, 0], [1, -1, 0], [0, 1, 0], [-1, -1, 1], [1, -1, 1], [0, 1, 1]], np.float32)
    a = oracle.OracleRenderer(v, np.array([[0, 1, 2], [3, 4, 5]], np.int32)).render(_camera())[0]
    b = oracle.OracleRenderer(v, np.array([[0, 2, 1], [5, 3, 4]], np.int32)).render(_camera())[0]   # no back-face culling
    np.testing.assert_array_equal(a, b)
    assert (a[32, 24] == 1)                                       # z=1 is nearer to the camera at z=5
    assert set(np.unique(a)) <= {0, 1, int(BG)}


def test_watertight_shared_edges_and_tie_break(oracle):
    # a fan of triangles around a vertex that projects exactly onto a pixel centre, edges through pixel centres
    from semantic_meshes_amd import data
    R = np.eye(3, dtype=np.float32)
    cam = data.Camera(R, np.zeros(3, np.float32), np.array([32, 32]), np.array([16.0, 16.0]), np.array([16.5, 16.5]))
    c = np.array([0, 0, 1.0])
    ring = [np.array([np.cos(a), np.sin(a), 1.0]) for a in np.linspace(0, 2 * np.pi, 9)[:-1]]
    v = np.array([c] + ring, np.float32)
    f = np.array([[0, 1 + i, 1 + (i + 1) % 8] for i in range(8)], np.int32)
    idx, _ = oracle.OracleRenderer(v, f).render(cam)
    inside = np.zeros((32, 32), bool)
    for x in range(32):
        for y in range(32):
            px, py = (x + 0.5 - 16.5) / 16.0, (y + 0.5 - 16.5) / 16.0
            inside[x, y] = px * px + py * py < 0.85 ** 2            # well inside the octagon
    assert (idx[inside] != BG).all()                              # no cracks on shared edges
    # two coincident triangles: equal depth everywhere -> the lower id wins (B-4)
    v2 = np.array([[-1, -1, 1], [1, -1, 1], [0, 1, 1]], np.float32)
    idx2, _ = oracle.OracleRenderer(v2, np.array([[0, 1, 2], [0, 1, 2]], np.int32)).render(cam)
    assert set(np.unique(idx2)) == {0, int(BG)}


def test_watertight_exactly_once_with_rounded_edge_functions(oracle):
    """The edge functions are fma-rounded linear forms, not exact: watertightness rests on both triangles of a shared edge computing
    the SAME value.  A jittered grid mesh (irrational-looking coordinates, also far outside the image) at constant depth: the
    per-triangle coverages -- every triangle rendered on its own -- must sum to exactly one wherever the union covers a pixel."""
    from semantic_meshes_amd import data
    rng = np.random.default_rng(42)
    for trial, (n, span, focal) in enumerate([(7, 1.0, 23.7), (5, 9.0, 11.3), (9, 0.6, 61.9)]):
        W = H = 48
        cam = data.Camera(np.eye(3, dtype=np.float32), np.zeros(3, np.float32), np.array([W, H]), np.array([focal, focal * 1.07]),
                          np.array([W / 2 + 0.37, H / 2 - 0.21]))
        g = np.linspace(-span, span, n)
        X, Y = np.meshgrid(g, g, indexing="ij")
        X = X + rng.uniform(-0.3, 0.3, X.shape) * (2 * span / (n - 1))
        Y = Y + rng.uniform(-0.3, 0.3, Y.shape) * (2 * span / (n - 1))
        v = np.stack([X.ravel(), Y.ravel(), np.full(X.size, 1.5)], axis=1).astype(np.float32)
        faces = []
        for i in range(n - 1):
            for j in range(n - 1):
                a, b, c, d = i * n + j, (i + 1) * n + j, i * n + j + 1, (i + 1) * n + j + 1
                faces += ([[a, b, c], [b, d, c]] if (i + j + trial) % 2 else [[a, b, d], [a, d, c]])   # both diagonals, both windings below
        f = np.array(faces, np.int32)
        flip = rng.random(len(f)) < 0.5
        f[flip] = f[flip][:, ::-1]                                  # orientation must not matter (no back-face culling)
        union, _ = oracle.OracleRenderer(v, f).render(cam)
        times = np.zeros((W, H), np.int32)
        for t in range(len(f)):
            one, _ = oracle.OracleRenderer(v, f[t:t + 1]).render(cam)
            times += (one != BG)
        assert ((union != BG) == (times > 0)).all()
        assert times.max() == 1, "a sample on a shared edge was claimed by both triangles"
        assert (times[union != BG] == 1).all()
        # interior of the mesh (away from its jittered outline): no cracks
        u = (v[:, 0] / 1.5) * focal + W / 2 + 0.37
        w = (v[:, 1] / 1.5) * focal * 1.07 + H / 2 - 0.21
        xs = np.arange(W) + 0.5; ys = np.arange(H) + 0.5
        inner = (xs[:, None] > u.reshape(n, n)[0, :].max() + 1) & (xs[:, None] < u.reshape(n, n)[-1, :].min() - 1) & \
                (ys[None, :] > w.reshape(n, n)[:, 0].max() + 1) & (ys[None, :] < w.reshape(n, n)[:, -1].min() - 1)
        assert (times[inner] == 1).all()
        assert trial == 1 or inner.sum() > 50       # (trial 1: the mesh is far larger than the image -- every pixel is interior)


def test_behind_camera_triangles_are_dropped(oracle):
    v = np.array([[-1, -1, 6], [1, -1, 6], [0, 1, 4]], np.float32)    # one vertex in front, two behind the camera at z=5
    idx, depth = oracle.OracleRenderer(v, np.array([[0, 1, 2]], np.int32)).render(_camera())
    assert (idx == BG).all()


def test_golden_cfg1_render(oracle):
    from semantic_meshes_amd import data
    g = np.load(os.path.join(GOLDEN, "cfg1_render.npz"))
    r = oracle.OracleRenderer(g["vertices"], g["faces"])
    assert r.getPrimitivesNum() == 10000
    for k in range(4):
        cam = data.Camera(g["cam%d_R" % k], g["cam%d_t" % k], g["cam%d_res" % k], g["cam%d_f" % k], g["cam%d_c" % k])
        idx, depth = r.render(cam)
        want = _unrle(g["idx%d_vals" % k], g["idx%d_lens" % k], tuple(g["cam%d_res" % k]))
        np.testing.assert_array_equal(idx, want)
        assert np.bitwise_xor.reduce(depth.view(np.uint32).reshape(-1)) == g["depth%d_xor" % k]


def test_golden_scene_matches_generator():
    """The committed fixture inputs are what synth.scene('cfg1') produces on this machine."""
    from semantic_meshes_amd import synth
    g = np.load(os.path.join(GOLDEN, "cfg1_render.npz"))
    mesh, cams, C = synth.scene("cfg1")
    np.testing.assert_array_equal(mesh.vertices, g["vertices"])
    np.testing.assert_array_equal(mesh.faces, g["faces"])
    assert C == 5 and len(cams) == 4


# ---- fusion ---------------------------------------------------------------------------------------
def _simple(oracle, kind="sum", iew=0.5, P=4, C=3):
    return oracle.OracleAggregator(P, C, kind, iew)


def test_ka4_out_of_range_indices_ignored(oracle):
    agg = _simple(oracle)
    probs = np.full((2, 2, 3), 1 / 3, np.float32)
    agg.add(np.array([[4, 0xFFFFFFFF], [7, 100]], np.uint32), probs)
    assert (agg.get_raw() == 0).all()
    agg.add(np.array([[-1, -1], [-1, 2]], np.int64), probs)         # int64 -1 -> 0xFFFFFFFF (Fusion.h:45)
    raw = agg.get_raw()
    assert (raw[:2] == 0).all() and (raw[3] == 0).all() and (raw[2] > 0).all()


def test_ka5_ka6_dont_care_counted_and_image_weights(oracle):
    idx = np.zeros((4, 2), np.uint32)
    probs = np.zeros((4, 2, 3), np.float32)
    probs[:2, :, 1] = 1.0                                           # 4 valid pixels, 4 don't-care (sum 0 <= 0.5)
    for iew, want in ((0.0, 4.0), (1.0, 0.5), (0.5, 2.25)):         # w = iew/8 + (1 - iew): count covers all 8 pixels
        agg = _simple(oracle, iew=iew)
        agg.add(idx, probs)
        np.testing.assert_allclose(agg.get_raw()[0], [0, want, 0], rtol=1e-6)
    agg = _simple(oracle, iew=0.0)
    agg.add(idx, probs, np.full((4, 2), 0.25, np.float32))          # per-pixel weights multiply (Mesh.h:103)
    np.testing.assert_allclose(agg.get_raw()[0], [0, 1.0, 0], rtol=1e-6)


def test_ka7_ka8_get_normalises_and_zeroes_untouched(oracle):
    agg = _simple(oracle)
    probs = np.zeros((1, 2, 3), np.float32)
    probs[0, 0] = [0.2, 0.5, 0.3]
    probs[0, 1] = [0.6, 0.3, 0.1]
    agg.add(np.array([[1, 1]], np.uint32), probs)
    out = agg.get()
    np.testing.assert_allclose(out[1], [0.4, 0.4, 0.2], rtol=1e-6)
    np.testing.assert_allclose(out[1].sum(), 1.0, rtol=1e-6)
    assert (out[[0, 2, 3]] == 0).all()                              # 0/0 -> NaN -> 0 (Fusion.h:79-95)
    agg.reset()
    assert (agg.get() == 0).all()


def test_ka11_summax_only_argmax_contributes(oracle):
    agg = _simple(oracle, "summax", 0.0)
    probs = np.zeros((1, 3, 3), np.float32)
    probs[0, 0] = [0.2, 0.5, 0.3]
    probs[0, 1] = [0.6, 0.3, 0.1]
    probs[0, 2] = [0.4, 0.4, 0.2]                                   # tie -> first max (Fusion.cu:53)
    agg.add(np.zeros((1, 3), np.uint32), probs)
    np.testing.assert_allclose(agg.get_raw()[0], [1.0, 0.5, 0.0], rtol=1e-6)


def test_ka12_mul_aggregator(oracle):
    agg = _simple(oracle, "mul", 0.0)
    probs = np.zeros((1, 2, 3), np.float32)
    probs[0, 0] = [0.2, 0.5, 0.3]
    probs[0, 1] = [0.5, 0.25, 0.25]
    agg.add(np.array([[2, 2]], np.uint32), probs)
    out = agg.get()
    want = np.array([0.1, 0.125, 0.075])
    np.testing.assert_allclose(out[2], want / want.sum(), rtol=1e-5)
    np.testing.assert_allclose(out[0], 1 / 3, rtol=1e-6)           # untouched Mul rows: exp(0) normalised (A.4)


def test_mul_spec_against_the_literal_pow_then_log_reading(oracle):
    """the spec'd Mul term w * log_spec(p) against an INDEPENDENT restatement of the literal reading of
    Fusion.cu:83-87 -- logf(powf(p, w)), p^w rounded to float32 before the log, libm on both calls.  (i) For probabilities a
    network emits the two agree to 1e-5 on get(); (ii) where p^w underflows float32 (w * ln p < ~-103) the literal form yields
    -inf and eliminates the class for good, the spec'd form keeps a finite term: the one known behavioural difference."""
    rng = np.random.default_rng(5)
    W, H, C, P = 48, 40, 6, 30
    idx = rng.integers(0, P, (W, H)).astype(np.uint32)
    oracle.set_accum_double(True)
    try:
        res = {}
        for literal in (False, True):
            oracle.set_mul_literal(literal)
            agg = oracle.OracleAggregator(P, C, "mul", 0.5)
            r2 = np.random.default_rng(6)
            for view in range(4):
                p = r2.random((W, H, C), dtype=np.float32) ** 2 + 1e-3
                p /= p.sum(axis=-1, keepdims=True)
                agg.add(idx, p, r2.random((W, H), dtype=np.float32) + 0.25)
            res[literal] = agg.get()
        np.testing.assert_allclose(res[False], res[True], rtol=1e-5, atol=1e-7)
        # (ii) the divergence, pinned: one pixel, p = (1e-30, 1 - 1e-30), weight 4: p^w = 1e-120 underflows float32
        out = {}
        for literal in (False, True):
            oracle.set_mul_literal(literal)
            agg = oracle.OracleAggregator(1, 2, "mul", 0.0)
            agg.add(np.zeros((1, 1), np.uint32), np.array([[[1e-30, 1.0]]], np.float32), np.full((1, 1), 4.0, np.float32))
            out[literal] = agg.get_raw()[0]
        assert np.isneginf(out[True][0]) and np.isfinite(out[False][0]) and abs(out[False][0] - 4 * np.log(1e-30)) < 1e-3
    finally:
        oracle.set_mul_literal(False)
        oracle.set_accum_double(False)


def test_edge_function_spec_against_the_five_operation_form(oracle):
    """the spec'd edge function fma(A, px, fma(B, py, C)) (two roundings) against the round-1 form
    sign * (dx (py - ly) - dy (px - lx)) (five), kept as an independent yardstick: they can only disagree on samples within a
    rounding error of an edge.  cfg1 views: identical index images up to a handful of pixels, depth where they agree to 1e-6."""
    from semantic_meshes_amd import synth
    mesh, cams, _ = synth.scene("cfg1")
    r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    for cam in cams[:2]:
        idx, depth = r.render(cam)
        oracle.set_edge_five_op(True)
        try:
            idx5, depth5 = r.render(cam)
        finally:
            oracle.set_edge_five_op(False)
        differ = idx != idx5
        assert differ.mean() < 1e-4, differ.sum()
        same = ~differ & (idx != BG)
        np.testing.assert_allclose(depth[same], depth5[same], rtol=1e-6)
        assert ((idx == BG) == (idx5 == BG)).mean() > 0.9999          # watertight either way: no cracks appear or vanish


def test_ka9_shape_errors(oracle):
    agg = _simple(oracle)
    with pytest.raises(ValueError):
        agg.add(np.zeros((2, 2), np.uint32), np.zeros((2, 3, 3), np.float32))
    with pytest.raises(ValueError):
        agg.add(np.zeros((2, 2), np.float32), np.zeros((2, 2, 3), np.float32))


def test_golden_cfg1_fuse(oracle):
    from semantic_meshes_amd import synth
    g = np.load(os.path.join(GOLDEN, "cfg1_render.npz"))
    want = np.load(os.path.join(GOLDEN, "cfg1_fuse.npz"))
    for kind in ("sum", "summax", "mul"):
        agg = oracle.OracleAggregator(10000, 5, kind, 0.5)
        for k in range(4):
            W, H = (int(v) for v in g["cam%d_res" % k])
            idx = _unrle(g["idx%d_vals" % k], g["idx%d_lens" % k], (W, H))
            agg.add(idx, oracle.synth_probs(W * H, 5, synth.probs_seed(7, k), 0.05).reshape(W, H, 5))
        np.testing.assert_array_equal(agg.get(), want[kind])
    # the Mul fixture of the float64-accumulating yardstick (what the HIP path's (hi, lo) state is held to, 1e-5): reproduced bit for
    # bit, and the float32 log-domain state the reference keeps (LogProb<float>, Fusion.cu:85) lies 1.5e-4 from it after four views
    oracle.set_accum_double(True)
    try:
        agg = oracle.OracleAggregator(10000, 5, "mul", 0.5)
        for k in range(4):
            W, H = (int(v) for v in g["cam%d_res" % k])
            idx = _unrle(g["idx%d_vals" % k], g["idx%d_lens" % k], (W, H))
            agg.add(idx, oracle.synth_probs(W * H, 5, synth.probs_seed(7, k), 0.05).reshape(W, H, 5))
        np.testing.assert_array_equal(agg.get(), want["mul_float64_state"])
    finally:
        oracle.set_accum_double(False)
    np.testing.assert_allclose(want["mul"], want["mul_float64_state"], rtol=2e-4, atol=1e-6)
    assert np.abs(want["mul"] - want["mul_float64_state"]).max() > 1e-5


def test_strided_and_threaded_add_agree(oracle):
    rng = np.random.default_rng(0)
    W, H, C, P = 40, 30, 4, 60
    idx = rng.integers(0, P, (W, H)).astype(np.int32)
    hwc = rng.random((H, W, C), dtype=np.float32)
    a1, a2 = oracle.OracleAggregator(P, C), oracle.OracleAggregator(P, C)
    a1.add(idx, np.ascontiguousarray(hwc.transpose(1, 0, 2)))
    oracle.set_threads(4)
    try:
        a2.add(idx, hwc.transpose(1, 0, 2))                         # strided view, OpenMP + per-primitive locks
    finally:
        oracle.set_threads(1)
    np.testing.assert_allclose(a1.get(), a2.get(), rtol=1e-5, atol=1e-7)


def test_mul_log_spec_is_within_an_ulp_of_ln(oracle):
    """Mul's logarithm is a fixed float32 operation sequence shared by the oracle and the HIP kernels (DESIGN.md
    3.3): pinned here against numpy's double-precision log."""
    import ctypes
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.random(500_000, dtype=np.float32), np.exp(rng.uniform(-100, 88, 500_000)).astype(np.float32),
                        np.float32([1e-45, 1e-40, 1.17549435e-38, 1.0, 2.0, 0.5, 3.4e38, 1.41421354, 1.41421366, 0.70710677,
                                    np.nextafter(np.float32(1), np.float32(2)), np.nextafter(np.float32(1), np.float32(0))])])
    x = x[x > 0]
    out = np.empty_like(x)
    oracle.lib().smesh_oracle_log_spec(x.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(len(x)))
    ref = np.log(x.astype(np.float64))
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    assert (np.abs(out.astype(np.float64) - ref) <= 1.0 * np.maximum(ulp, 1e-300)).all()
    special = np.float32([0.0, -1.0, np.inf, np.nan])
    got = np.empty_like(special)
    oracle.lib().smesh_oracle_log_spec(special.ctypes.data_as(ctypes.c_void_p), got.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(4))
    assert got[0] == -np.inf and np.isnan(got[1]) and got[2] == np.inf and np.isnan(got[3])


def test_texel_counts(oracle):
    # KA10: r(r+1)/2 texels per triangle (TexturedTriangleRenderer.h:43-47)
    v = np.array([[0.4, 0, 0], [0.5, 1, 0], [0.6, 0, 0]], np.float32)   # debug_render_texels.py:19-23
    cam = _camera(200, 200, eye=(0.5, 0.5, 2.0), target=(0.5, 0.5, 0), f=180.0)
    r = oracle.OracleRenderer(v, np.array([[0, 1, 2]], np.int32), [cam], 0.1)
    faces, res, first = r.texel_layout()
    assert r.getPrimitivesNum() == int(res[0]) * (int(res[0]) + 1) // 2 and first[0] == 0 and res[0] >= 2
    idx, _ = r.render(cam)
    seen = np.unique(idx[idx != BG])
    assert seen.max() < r.getPrimitivesNum() and len(seen) >= r.getPrimitivesNum() // 2


def test_annotation_renderer_gathers_rows(oracle):
    # ModelRenderer::render (Mesh.h:25-42): annotation row under each pixel, background where idx >= P
    agg = _simple(oracle, P=3, C=2)
    probs = np.zeros((1, 2, 2), np.float32)
    probs[0, 0] = [0.25, 0.75]
    probs[0, 1] = [1.0, 0.0]
    agg.add(np.array([[0, 2]], np.uint32), probs)
    idx = np.array([[0, 1, 2], [7, 0xFFFFFFFF, 2]], np.uint32)
    img = oracle.render_annotations(agg, idx, [9.0, 8.0])
    np.testing.assert_allclose(img[0], [[0.25, 0.75], [0, 0], [1, 0]], rtol=1e-6)
    np.testing.assert_allclose(img[1], [[9, 8], [9, 8], [1, 0]], rtol=1e-6)


def test_fast_histogram_variant_is_equivalent(oracle):
    """bench.py's "optimised_cpu" leg swaps the reference's serial std::map histogram (Mesh.h:90-93) for a dense
    parallel one: the counts, hence the fused accumulators, must be bit-identical."""
    from helpers import random_probs, small_scene
    mesh, cams = small_scene()
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    P, C = len(mesh.faces), 5
    rng = np.random.default_rng(0)
    a1, a2 = oracle.OracleAggregator(P, C, "summax", 0.7), oracle.OracleAggregator(P, C, "summax", 0.7)
    try:
        for cam in cams:
            probs = random_probs(rng, *cam.resolution, C)
            idx = o.render(cam)[0]
            oracle.set_fast_histogram(False)
            a1.add(idx, probs)
            oracle.set_fast_histogram(True)
            a2.add(idx, probs)
    finally:
        oracle.set_fast_histogram(False)
    np.testing.assert_array_equal(a1.get_raw().view(np.uint32), a2.get_raw().view(np.uint32))
