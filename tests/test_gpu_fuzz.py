"""Randomised differential test of the HIP path against the CPU oracle: triangle soups (overlapping, intersecting,
degenerate and partly behind-the-camera triangles of every size class), random cameras, resolutions, class counts and
aggregators.  Indices and depth must be bit-equal; fused distributions within the float tolerance."""
import numpy as np
import pytest

from helpers import assert_fused_close, random_probs

pytestmark = pytest.mark.gpu


def _soup(rng, nverts, nfaces, spread):
    """Random triangles: each face picks a centre and three vertices within `spread` of it (so sizes vary with spread)."""
    centres = rng.uniform(-1.0, 1.0, (nfaces, 3)).astype(np.float32)
    verts = (centres[:, None, :] + rng.normal(0.0, spread, (nfaces, 3, 3))).astype(np.float32).reshape(-1, 3)
    faces = np.arange(3 * nfaces, dtype=np.int32).reshape(nfaces, 3)
    # share some vertices between faces (watertight edges are where the tie rule matters)
    share = rng.integers(0, 3 * nfaces, nfaces // 2)
    faces.reshape(-1)[rng.integers(0, 3 * nfaces, nfaces // 2)] = share
    # a few exactly duplicated faces and zero-area faces
    faces[rng.integers(0, nfaces, 3)] = faces[rng.integers(0, nfaces, 3)]
    z = rng.integers(0, nfaces, 3)
    faces[z, 2] = faces[z, 1]
    return verts, faces


def _camera(sm, rng, W, H):
    from semantic_meshes_amd import synth
    eye = rng.uniform(-2.5, 2.5, 3)
    if rng.random() < 0.3:
        eye *= 0.2                                    # inside the soup: triangles cross the near limit
    target = rng.uniform(-0.5, 0.5, 3)
    R, t = synth.look_at(tuple(eye), tuple(target), up=(0, 0, 1))
    f = float(rng.uniform(0.4, 1.5)) * W
    return sm.data.Camera(R, t, np.array([W, H]), np.array([f, f * rng.uniform(0.8, 1.2)]),
                          np.array([W * rng.uniform(0.3, 0.7), H * rng.uniform(0.3, 0.7)]))


@pytest.mark.parametrize("seed", range(60))
def test_random_soup_against_oracle(sm, oracle, seed):
    import types
    rng = np.random.default_rng(1000 + seed)
    nfaces = int(rng.choice([50, 400, 3000, 6000]))
    spread = float(rng.choice([0.004, 0.02, 0.08, 0.4]))            # sub-pixel ... larger than 64 x 64 boxes
    verts, faces = _soup(rng, 0, nfaces, spread)
    W, H = int(rng.choice([37, 160, 333])), int(rng.choice([29, 120, 257]))
    C = int(rng.choice([1, 3, 5, 19, 21, 40, 47, 70, 130]))
    kind = str(rng.choice(["sum", "summax", "mul"]))
    iew = float(rng.choice([0.0, 0.5, 1.0]))
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(verts, faces)
    P = len(faces)
    agg = sm.fusion.MeshAggregator(P, C, kind, iew)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind, iew)
        for view in range(3):
            cam = _camera(sm, rng, W, H)
            idx, depth = r.render(cam)
            oidx, odepth = o.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oidx)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
            probs = random_probs(rng, W, H, C, zero_fraction=0.1)
            if kind == "mul":
                probs = np.where(probs.sum(-1, keepdims=True) > 0, np.maximum(probs, 1e-3), 0).astype(np.float32)
            weights = rng.random((W, H), dtype=np.float32) if view == 1 else None
            if view == 2:
                agg.add(idx, probs)                                   # render() + add(): same kernels as fuse_view
            else:
                agg.fuse_view(r, cam, probs, weights)
            oagg.add(oidx, probs, weights)
        # Mul: (hi, lo) rows in every triangle-order kernel, a view's terms summed in double (fuse_tri.inc.hpp, "Mul state")
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("seed", list(range(12)) + [743])   # 743: a texel resolution one ulp of area away from the next integer
def test_random_texel_soup_against_oracle(sm, oracle, seed):
    import types
    rng = np.random.default_rng(5000 + seed)
    nfaces = int(rng.choice([60, 500, 2500]))
    spread = float(rng.choice([0.01, 0.05, 0.3]))
    verts, faces = _soup(rng, 0, nfaces, spread)
    W, H = int(rng.choice([64, 200])), int(rng.choice([48, 150]))
    C = int(rng.choice([2, 7, 40, 45]))
    kind = str(rng.choice(["sum", "summax"]))
    cams = [_camera(sm, rng, W, H) for _ in range(3)]
    tpp = float(rng.choice([0.1, 0.5, 1.5]))
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    r = sm.render.texels(mesh, cams, tpp)
    o = oracle.OracleRenderer(verts, faces, cams, tpp)
    P = r.getPrimitivesNum()
    assert P == o.getPrimitivesNum()
    if P == 0:
        return
    agg = sm.fusion.MeshAggregator(P, C, kind)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        batch = []
        for cam in cams:
            idx, depth = r.render(cam)
            oidx, odepth = o.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oidx)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
            probs = random_probs(rng, W, H, C, zero_fraction=0.1)
            if seed % 2:
                batch.append(probs)                                   # odd seeds: all views in one fuse_views call (grouped rasteriser launches)
            else:
                agg.fuse_view(r, cam, probs)
            oagg.add(oidx, probs)
        if batch:
            from semantic_meshes_amd.device import to_device
            agg.fuse_views(r, cams, [to_device(p) for p in batch])
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("seed", range(6))
def test_random_soup_forced_reorder(seed):
    """The same differential test with SMESH_REORDER=1 (every mesh processed in Morton order behind a position -> id table)."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, SMESH_REORDER="1")
    here = os.path.dirname(os.path.abspath(__file__))
    sel = "test_random_soup_against_oracle and (%s)" % " or ".join("[%d]" % (seed * 10 + k) for k in range(10))
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_fuzz.py"), "-q", "-x", "-m", "gpu",
                          "-k", sel, "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]


@pytest.mark.parametrize("seed", range(40))
def test_random_soup_fuse_views_against_oracle(sm, oracle, seed):
    """fuse_views on soups: the two views of a pair see every triangle at unrelated sizes (small in one, big or huge in the
    other, culled in either), at different resolutions, with and without weights; device-resident images (pairs) and odd numbers
    of views."""
    import types
    from semantic_meshes_amd.device import to_device
    rng = np.random.default_rng(9000 + seed)
    nfaces = int(rng.choice([50, 400, 3000, 6000]))
    spread = float(rng.choice([0.004, 0.02, 0.08, 0.4]))
    verts, faces = _soup(rng, 0, nfaces, spread)
    C = int(rng.choice([1, 3, 5, 19, 21, 40, 47, 48]))                # all within k_fuse_tri: pairs
    kind = str(rng.choice(["sum", "summax", "mul"]))
    iew = float(rng.choice([0.0, 0.5, 1.0]))
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(verts, faces)
    P = len(faces)
    nviews = int(rng.choice([2, 3, 5, 8, 9]))                         # up to five pairs in one call
    cams, probs, weights = [], [], []
    for v in range(nviews):
        W, H = int(rng.choice([37, 160, 333])), int(rng.choice([29, 120, 257]))
        cams.append(_camera(sm, rng, W, H))
        p = random_probs(rng, W, H, C, zero_fraction=0.1)
        if kind == "mul":
            p = np.where(p.sum(-1, keepdims=True) > 0, np.maximum(p, 1e-3), 0).astype(np.float32)
        probs.append(p)
        weights.append(rng.random((W, H), dtype=np.float32) if rng.random() < 0.5 else None)
    use_w = any(w is not None for w in weights)
    agg = sm.fusion.MeshAggregator(P, C, kind, iew)
    agg.fuse_views(r, cams, [to_device(p) for p in probs],
                   [None if w is None else to_device(w) for w in weights] if use_w else None)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind, iew)
        for v in range(nviews):
            oagg.add(o.render(cams[v])[0], probs[v], weights[v])
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)


@pytest.mark.parametrize("seed", [252362, 253169, 553071, 702926])
def test_soup_fuse_views_seeds_that_caught_the_per_view_plane_decision(sm, oracle, seed):
    """Round 6 let a view of fuse_views skip its index plane when it had no queued triangles OF ITS OWN -- but a triangle that is big in
    one view of a launch has its small views scanned in THEIR planes too (fuse_tri.inc.hpp, the tail waves).  Two of 7 500 random soups of the
    round's differential sweep (fifty triangles of a few pixels, several views) showed it: 20 of 150 elements 1 % off.  The decision is
    now one per raster launch wherever a fusion kernel does that (raster.hip: view_needs_planes, level 1), and each view's own where every
    launch is k_fuse_tri, which reads such views from their records' masks (level 2; fuse_box by_mask).  Seeds 553071 and 702926 (47 / 48
    classes, eight / nine views: k_fuse_tri_any, not k_fuse_tri as the kernel's NAME for that class count says) caught the first level 2."""
    test_random_soup_fuse_views_against_oracle(sm, oracle, seed)


@pytest.mark.parametrize("seed", range(14))
def test_random_texel_soup_mul_against_the_float64_oracle(sm, oracle, seed):
    """Mul on texel renderers, triangles of every size: small ones fold each term into the (hi, lo) row in double (k_fuse_texel /
    k_fuse_texel_multi), the coarse texels of big triangles -- thousands of pixels each -- sum a view's terms in double scratch rows
    before they are folded (k_fuse_texel_big).  Same bound as Sum / Summax."""
    import types
    from semantic_meshes_amd.device import to_device
    rng = np.random.default_rng(7000 + seed)
    nfaces = int(rng.choice([60, 500, 2500]))
    spread = float(rng.choice([0.02, 0.08, 0.4]))
    verts, faces = _soup(rng, 0, nfaces, spread)
    W, H = int(rng.choice([64, 200, 333])), int(rng.choice([48, 150]))
    C = int(rng.choice([2, 7, 19, 40]))
    iew = float(rng.choice([0.0, 0.5, 1.0]))
    cams = [_camera(sm, rng, W, H) for _ in range(int(rng.choice([1, 3, 9])))]
    tpp = float(rng.choice([0.1, 0.5, 1.5]))
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    r = sm.render.texels(mesh, cams, tpp)
    o = oracle.OracleRenderer(verts, faces, cams, tpp)
    P = r.getPrimitivesNum()
    if P == 0:
        return
    probs = []
    for cam in cams:
        p = random_probs(rng, W, H, C, zero_fraction=0.1)
        probs.append(np.where(p.sum(-1, keepdims=True) > 0, np.maximum(p, 1e-3), 0).astype(np.float32))
    agg = sm.fusion.MeshAggregator(P, C, "mul", iew)
    if seed % 2:
        agg.fuse_views(r, cams, [to_device(p) for p in probs])
    else:
        for cam, p in zip(cams, probs):
            agg.fuse_view(r, cam, p)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, "mul", iew)
        for cam, p in zip(cams, probs):
            oagg.add(o.render(cam)[0], p)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)


def _room(rng, n):
    """A closed box whose six walls are n x n jittered quads (2 n^2 triangles each): seen from inside, the walls beside and behind
    the camera cross the camera plane (near-plane clipping, raster spec 1b), and with small n the triangles are medium-sized."""
    half = rng.uniform(1.0, 3.0, 3)
    verts, faces = [], []
    for axis in range(3):
        for side in (-1.0, 1.0):
            u, v = [a for a in range(3) if a != axis]
            base = len(verts)
            g = np.linspace(-1.0, 1.0, n + 1)
            for i in range(n + 1):
                for j in range(n + 1):
                    p = np.zeros(3)
                    p[axis] = side * half[axis]
                    p[u], p[v] = g[i] * half[u], g[j] * half[v]
                    verts.append(p)
            for i in range(n):
                for j in range(n):
                    a, b, c, d = base + i * (n + 1) + j, base + (i + 1) * (n + 1) + j, base + (i + 1) * (n + 1) + j + 1, base + i * (n + 1) + j + 1
                    faces += [(a, b, c), (a, c, d)] if (i + j) % 2 else [(a, b, d), (b, c, d)]
    verts = np.asarray(verts)
    # interior vertices of a wall move a little inside their wall's plane (the room stays closed: borders are shared and fixed)
    return verts.astype(np.float32), np.asarray(faces, np.int32), half


@pytest.mark.parametrize("seed", range(24))
def test_random_room_against_oracle(sm, oracle, seed):
    """Cameras INSIDE closed rooms (eval-scannet/eval_scannet.py:203-238's regime): clipped walls, medium and large triangles, every
    aggregator; indices and depth bit-equal, no background pixel, fused distributions against the float64 oracle -- single views and
    fuse_views groups."""
    import types
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.device import to_device
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([1, 3, 8, 20]))
    verts, faces, half = _room(rng, n)
    W, H = int(rng.choice([96, 320, 640])), int(rng.choice([72, 240, 480]))
    C = int(rng.choice([3, 19, 40, 64]))
    kind = str(rng.choice(["sum", "summax", "mul"]))
    r = sm.render.triangles(types.SimpleNamespace(vertices=verts, faces=faces))
    o = oracle.OracleRenderer(verts, faces)
    P = len(faces)
    agg, grouped = sm.fusion.MeshAggregator(P, C, kind), sm.fusion.MeshAggregator(P, C, kind)
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        cams, images = [], []
        for view in range(3):
            eye = rng.uniform(-0.85, 0.85, 3) * half
            target = rng.uniform(-1.0, 1.0, 3) * half
            R, t = synth.look_at(tuple(eye), tuple(target), up=(0, 0, 1))
            f = float(rng.uniform(0.35, 1.2)) * W
            cam = sm.data.Camera(R, t, np.array([W, H]), np.array([f, f]), np.array([W / 2.0, H / 2.0]))
            idx, depth = r.render(cam)
            oidx, odepth = o.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oidx)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), odepth.view(np.uint32))
            assert (oidx != 0xFFFFFFFF).all()                          # a closed room: no pixel sees the background
            probs = random_probs(rng, W, H, C, zero_fraction=0.1)
            if kind == "mul":
                probs = np.where(probs.sum(-1, keepdims=True) > 0, np.maximum(probs, 1e-3), 0).astype(np.float32)
            agg.fuse_view(r, cam, probs)
            oagg.add(oidx, probs)
            cams.append(cam)
            images.append(to_device(probs))
        grouped.fuse_views(r, cams, images)
        want = oagg.get()
        assert_fused_close(agg.get(), want, rtol=1e-5, atol=1e-6)
        assert_fused_close(grouped.get(), want, rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)


def _relief_mesh(rng, a, b):
    """A jittered height field of 2 * a * b triangles (shared vertices: watertight edges), some of it folded over itself."""
    from semantic_meshes_amd import synth
    mesh = synth.grid_mesh(a, b, relief=float(rng.choice([0.15, 0.6, 2.0])))
    v = mesh.vertices.copy()
    s = 10.0 / a
    v[:, :2] += rng.normal(0.0, 0.3 * s, (len(v), 2)).astype(np.float32)      # irregular triangles, some slivers
    v[:, 2] += rng.normal(0.0, float(rng.choice([0.0, 0.5, 3.0])) * s, len(v)).astype(np.float32)
    return v.astype(np.float32), mesh.faces.copy()


@pytest.mark.parametrize("seed", range(8))
def test_random_dense_mesh_of_medium_triangles_against_oracle(sm, oracle, seed):
    """Meshes of 33 000 - 100 000 triangles whose boxes are mostly 9 - 24 pixels wide (the decimated-scan regime,
    eval-scannet/simplify_scannet_meshes.py:65-82): the rasteriser's waves hold 32 or 64 triangles and a lane walks its own box sub-box
    by sub-box (raster.hip raster_frag_64), next to boxes beyond 24 pixels in the same wave (cooperative loop), triangles across the
    near plane (cameras close to the surface) and huge ones.  Indices and depth bit-equal for triangle and texel renderers; fuse_views
    (the views of a group in one rasteriser launch) against the float64 oracle."""
    import types
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.device import to_device
    rng = np.random.default_rng(77000 + seed)
    a, b = [(130, 128), (150, 110), (200, 100), (180, 120), (260, 130)][int(rng.integers(0, 5))]
    verts, faces = _relief_mesh(rng, a, b)
    rs = float(rng.uniform(0.5, 0.8))                 # the ring's radius; a quad then measures ~ 0.8 W / (a * rs) pixels
    W = int(a * rs * float(rng.uniform(10.0, 18.0)) / 0.8)
    H = int(W * float(rng.uniform(0.55, 0.8)))
    C = int(rng.choice([5, 19, 40]))
    kind = str(rng.choice(["sum", "summax", "mul"]))
    mesh = types.SimpleNamespace(vertices=verts, faces=faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(verts, faces)
    oracle.set_threads(8)
    P = len(faces)
    cams = []
    for view in range(3):
        if view == 2 and rng.random() < 0.5:      # close to the surface, looking along it: everything from sub-pixel to clipped
            eye = (float(rng.uniform(-4, 4)), float(rng.uniform(-2, 2)), float(rng.uniform(0.3, 1.0)))
            target = (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5)), 0.0)
            R, t = synth.look_at(eye, target, up=(0, 0, 1))
            f = float(rng.uniform(0.5, 1.0)) * W
            cams.append(sm.data.Camera(R, t, np.array([W, H]), np.array([f, f]), np.array([W / 2.0, H / 2.0])))
        else:
            cams.append(synth.ring_camera(int(rng.integers(0, 40)), 40, W, H, radius_scale=rs))
    oidx = []
    medium = 0
    for cam in cams:
        idx, depth = r.render(cam)
        oi, od = o.render(cam)
        np.testing.assert_array_equal(np.asarray(idx), oi)
        np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), od.view(np.uint32))
        oidx.append(oi)
        medium = max(medium, r.render_stats(cam)[1][0])
    assert medium > P // 4, "the scene was meant to consist of boxes over 8 x 8 pixels"
    if seed % 2 == 0:       # the texel renderer over the same geometry (texel ids from the sub-box walk's barycentrics)
        rt = sm.render.texels(mesh, cams, texels_per_pixel=0.2)
        ot = oracle.OracleRenderer(verts, faces, cameras=cams, texels_per_pixel=0.2)
        for cam in cams[:2]:
            idx, depth = rt.render(cam)
            oi, od = ot.render(cam)
            np.testing.assert_array_equal(np.asarray(idx), oi)
            np.testing.assert_array_equal(np.asarray(depth).view(np.uint32), od.view(np.uint32))
    probs = [random_probs(rng, W, H, C, zero_fraction=0.1) for _ in cams]
    if kind == "mul":
        probs = [np.where(p.sum(-1, keepdims=True) > 0, np.maximum(p, 1e-3), 0).astype(np.float32) for p in probs]
    agg = sm.fusion.MeshAggregator(P, C, kind)
    agg.fuse_views(r, cams, [to_device(p) for p in probs])
    oracle.set_accum_double(True)
    try:
        oagg = oracle.OracleAggregator(P, C, kind)
        for oi, p in zip(oidx, probs):
            oagg.add(oi, p)
        assert_fused_close(agg.get(), oagg.get(), rtol=1e-5, atol=1e-6)
    finally:
        oracle.set_accum_double(False)
        oracle.set_threads(1)
