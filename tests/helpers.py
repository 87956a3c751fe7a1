"""Shared helpers of the parity tests."""
import numpy as np

BG = np.uint32(0xFFFFFFFF)


def small_scene(a=40, b=20, width=160, height=120, views=3):
    from semantic_meshes_amd import synth
    mesh = synth.grid_mesh(a, b)
    cams = [synth.ring_camera(k, views, width, height) for k in range(views)]
    return mesh, cams


def random_probs(rng, W, H, C, zero_fraction=0.05):
    p = rng.random((W, H, C), dtype=np.float32) ** 4 + 1e-4
    p /= p.sum(axis=-1, keepdims=True)
    zero = rng.random((W, H)) < zero_fraction
    p[zero] = 0.0
    return p.astype(np.float32)


def assert_fused_close(got, want, rtol=1e-5, atol=2e-7):
    """north_star tolerance: 1e-5 relative on the float aggregator (plus a tiny absolute floor for
    classes whose normalised probability is ~0)."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        need = ((err - atol) / np.maximum(np.abs(want), 1e-300))[bad].max()   # the rtol that would have passed
        raise AssertionError("%d / %d elements out of tolerance (rtol %.1e would need %.2e), worst abs err %.3e (want %.6g got %.6g)" % (
            bad.sum(), bad.size, rtol, need, err.max(), want.flat[err.argmax()], got.flat[err.argmax()]))


def raycast(cam, vertices, faces):
    """An INDEPENDENT geometric check of a render (numpy, float64): for every pixel centre the ray through it is intersected with
    every triangle in camera space (no projection of vertices, so triangles that cross the camera plane need no clipping).
    Returns (idx uint32 (W,H), depth float64 (W,H), b1, b2, margin): nearest hit with z_c > 1e-6, its camera-space z, the hit's
    barycentric coordinates in its triangle, and `margin` = how far (in barycentric units, and relative depth to the runner-up)
    the decision is from flipping -- pixels with a small margin lie on an edge / a depth tie and are excluded from comparisons."""
    W, H = cam.resolution
    R = np.asarray(cam.rotation, np.float32)
    t = np.asarray(cam.translation, np.float32)
    v = np.asarray(vertices, np.float32)
    vc = ((R[None, :, 0] * v[:, 0:1] + R[None, :, 1] * v[:, 1:2]) + R[None, :, 2] * v[:, 2:3]) + t[None, :]   # float32, as the spec
    vc = vc.astype(np.float64)
    fx, fy = cam.focal_lengths
    cx, cy = cam.principal_point
    xs, ys = np.meshgrid(np.arange(W) + 0.5, np.arange(H) + 0.5, indexing="ij")
    d = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], axis=-1)          # ray direction, z component 1: t == z_c
    best_z = np.full((W, H), np.inf)
    second_z = np.full((W, H), np.inf)
    best = np.full((W, H), 0xFFFFFFFF, np.uint32)
    bb1, bb2 = np.zeros((W, H)), np.zeros((W, H))
    edge_margin = np.full((W, H), np.inf)
    near_miss = np.full((W, H), np.inf)     # smallest |barycentric violation| among triangles that just missed, nearer than the hit
    for f, (i0, i1, i2) in enumerate(np.asarray(faces)):
        a, b, c = vc[i0], vc[i1], vc[i2]
        e1, e2 = b - a, c - a
        p = np.cross(d, e2)
        det = p @ e1
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            u = (p @ (-a)) * inv
            q = np.cross(-a, e1)
            w = (d @ q) * inv
            z = (q @ e2) * inv
        m = np.minimum(np.minimum(u, w), 1.0 - u - w)        # > 0 inside
        ok = np.isfinite(z) & (z > 1e-6) & np.isfinite(m)
        hit = ok & (m > 0)
        nm = ok & ~hit
        near_miss = np.where(nm & (z < best_z), np.minimum(near_miss, -m), near_miss)
        closer = hit & (z < best_z)
        second_z = np.where(closer, best_z, np.where(hit & (z < second_z), z, second_z))
        best_z = np.where(closer, z, best_z)
        best = np.where(closer, np.uint32(f), best)
        bb1, bb2 = np.where(closer, u, bb1), np.where(closer, w, bb2)
        edge_margin = np.where(closer, m, edge_margin)
    with np.errstate(invalid="ignore", divide="ignore"):
        depth_margin = np.where(np.isfinite(second_z), (second_z - best_z) / best_z, np.inf)
    # (near misses recorded before a nearer hit was found may belong to farther triangles: conservative, only shrinks the margin)
    margin = np.minimum(np.minimum(np.where(np.isfinite(best_z), edge_margin, np.inf), depth_margin), near_miss)
    return best, best_z, bb1, bb2, margin
