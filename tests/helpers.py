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
