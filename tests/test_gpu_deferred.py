"""Deferred grouping behind the per-view API (round 6).

The reference's loop is one `render(camera)` and one `add(indices, probs)` per view (python/scripts/colorize_cityscapes_mesh.py:54-67).
Here `render()` hands out planes that are rasterised on first use, and `add()` of such a plane -- like `fuse_view()` -- with class
vectors in the library's own device arrays joins a group of up to eight views that reaches the library as one `smesh_fuse_views`
call.  These tests pin what that may NOT change: the sums (bit for bit for Sum / Summax), the order of the views, what a plane
contains when somebody does look at it, and every point at which the deferred views must have been handed over.
"""
import queue
import threading

import numpy as np
import pytest

from helpers import small_scene, assert_fused_close

pytestmark = pytest.mark.gpu


def _scene(sm, views=11, C=19, seed=3):
    from semantic_meshes_amd import synth
    mesh, cams = small_scene(170, 81, 320, 240, views=views)     # (two- and three-pixel triangles: no float atomics, one order of additions)
    probs = [synth.device_probs(320, 240, C, synth.probs_seed(seed, k), 0.05, 0) for k in range(len(cams))]
    return mesh, cams, probs


@pytest.mark.parametrize("kind", ["sum", "summax", "mul"])
def test_deferred_render_add_equals_the_per_call_path(sm, kind):
    """render() + add() view by view with the grouping on and off, and fuse_view() likewise: raw accumulators bit-equal for Sum and
    Summax (per row the same float32 additions in the same order), within 1e-6 for Mul (its (hi, lo) pairs fold once per view either way)."""
    mesh, cams, probs = _scene(sm)
    P, C = len(mesh.faces), 19
    r = sm.render.triangles(mesh)
    raws, gets = {}, {}
    for mode in ("per-call", "deferred", "per-call fuse_view", "deferred fuse_view"):
        agg = sm.fusion.MeshAggregator(P, C, kind)
        agg.defer = mode.startswith("deferred")
        for k, cam in enumerate(cams):
            if "fuse_view" in mode:
                agg.fuse_view(r, cam, probs[k])
            else:
                idx, depth = r.render(cam)
                assert idx.unrun                                   # nothing has been rasterised for this call yet
                agg.add(idx, probs[k])
                assert idx.unrun == agg.defer                      # per-call: add() looked at the plane; deferred: nobody did
            if agg.defer:
                assert len(agg._pending) == (k + 1) % 8            # a group goes to the library on its eighth view
        raws[mode], gets[mode] = agg.get_raw(), agg.get()
        assert not agg._pending
    for mode in raws:
        if kind == "mul":
            assert_fused_close(gets[mode], gets["per-call"], rtol=1e-6)
        else:
            assert np.array_equal(raws[mode], raws["per-call"]), mode
            assert np.array_equal(gets[mode], gets["per-call"]), mode
    assert (gets["per-call"].sum(axis=1) > 0.5).sum() > P // 4


def test_a_lazy_plane_holds_the_render_whenever_it_is_looked_at(sm, oracle):
    """Looking at a plane (np.asarray, .ptr, DLPack, the array interface) rasterises it -- before or after the view went into a deferred
    group, in any order -- and the content is the oracle's."""
    mesh, cams, probs = _scene(sm, views=4)
    P = len(mesh.faces)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    want = [o.render(cam) for cam in cams]
    planes = [r.render(cam) for cam in cams]
    assert all(i.unrun and d.unrun for i, d in planes)
    agg = sm.fusion.MeshAggregator(P, 19)
    agg.add(planes[1][0], probs[1])                                    # deferred: still nothing rasterised
    assert planes[1][0].unrun and len(agg._pending) == 1
    for k in (2, 0, 3, 1):                                             # not in render order, depth before indices for one of them
        i, d = planes[k]
        if k == 3:
            assert np.array_equal(np.asarray(d).view(np.uint32), want[k][1].view(np.uint32))
        assert np.array_equal(np.asarray(i), want[k][0]) and not i.unrun and not d.unrun
        assert np.array_equal(np.asarray(d).view(np.uint32), want[k][1].view(np.uint32))
    assert planes[0][0].ptr != 0 and planes[0][0].__cuda_array_interface__["data"][0] == planes[0][0].ptr
    # a plane somebody looked at takes the per-call path (its records are on the device): same sums as the deferred view
    ref = sm.fusion.MeshAggregator(P, 19)
    ref.defer = False
    ref.add(r.render(cams[1])[0], probs[1])
    assert np.array_equal(agg.get_raw(), ref.get_raw())
    # dropping one plane of a pair before the other is looked at leaves nothing behind
    i, d = r.render(cams[0])
    del d
    assert np.array_equal(np.asarray(i), want[0][0])
    i, d = r.render(cams[0], lazy=False)
    assert not i.unrun and np.array_equal(np.asarray(i), want[0][0])


def test_deferred_views_keep_their_place_among_other_calls(sm):
    """Views that cannot wait -- host class vectors, an exported (foreign-writable) device array, fuse_views, add() of a host image --
    go to the library at once, BEHIND the deferred views before them: the order of the views is the caller's.  get(), get_raw(),
    reset(), _lib.synchronize() and anything else that uses the aggregator hand the group over first."""
    from semantic_meshes_amd import _lib
    mesh, cams, probs = _scene(sm, views=9)
    P, C = len(mesh.faces), 19
    r = sm.render.triangles(mesh)
    host3 = np.asarray(probs[3])
    exported = probs[5]
    _ = exported.__cuda_array_interface__                     # handed to another framework: it may be written behind our back
    plane7 = np.asarray(r.render(cams[7])[0])                 # a host copy of a render: the generic add() path

    def run(defer):
        agg = sm.fusion.MeshAggregator(P, C)
        agg.defer = defer
        for k in (0, 1, 2):
            agg.add(r.render(cams[k])[0], probs[k])
        assert len(agg._pending) == (3 if defer else 0)
        agg.add(r.render(cams[3])[0], host3)                  # host class vectors: consumed by the call
        assert not agg._pending
        agg.fuse_view(r, cams[4], probs[4])
        agg.add(r.render(cams[5])[0], exported)
        assert not agg._pending
        agg.fuse_view(r, cams[6], probs[6])
        agg.add(plane7, probs[7])
        assert not agg._pending
        agg.fuse_view(r, cams[8], probs[8])
        agg.fuse_views(r, cams[:2], probs[:2])
        assert not agg._pending
        return agg.get_raw()

    assert np.array_equal(run(True), run(False))
    agg = sm.fusion.MeshAggregator(P, C)
    agg.fuse_view(r, cams[0], probs[0])
    assert len(agg._pending) == 1
    _lib.synchronize(0)
    assert not agg._pending
    one = agg.get_raw()
    agg.fuse_view(r, cams[1], probs[1])
    agg.reset()                                               # the deferred view's sums would be cleared anyway
    assert not agg._pending and not agg.get_raw().any()
    agg.fuse_view(r, cams[0], probs[0])
    assert len(agg._pending) == 1
    got = agg.get()                                           # (get() hands the view over)
    assert not agg._pending and (got.sum(axis=1) > 0.5).any()
    assert np.array_equal(agg.get_raw(), one)
    # a group holds one renderer and one image size: another renderer closes it
    r2 = sm.render.triangles(mesh)
    agg.reset()
    agg.fuse_view(r, cams[0], probs[0])
    agg.fuse_view(r2, cams[1], probs[1])
    assert len(agg._pending) == 1 and agg._pending[0][0] is r2
    agg.fuse_view(r2, cams[2], probs[2])
    both = agg.get_raw()
    ref = sm.fusion.MeshAggregator(P, C)
    ref.fuse_views(r, cams[:3], probs[:3])
    assert np.array_equal(both, ref.get_raw())


def test_two_thread_harness_with_device_class_vectors(sm):
    """eval_scannet.py:189-238 with the network's output resident in HBM: the main thread renders view k + 1 (lazily: no device work)
    while the worker adds view k (deferred: a group of eight per library call).  Bit-equal to the same views fused by one thread
    without deferral, no deadlock."""
    mesh, cams, probs = _scene(sm, views=6)
    cams, probs = cams * 5, probs * 5
    P, C = len(mesh.faces), 19
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
                agg.add(*item)
        except Exception as e:  # pragma: no cover
            errors.append(e)

    t = threading.Thread(target=worker)
    t.start()
    for cam, p in zip(cams, probs):
        idx, depth = r.render(cam)
        q.put((idx, p))
    q.put(None)
    t.join(timeout=120)
    assert not t.is_alive() and not errors
    ref = sm.fusion.MeshAggregator(P, C)
    ref.defer = False
    for cam, p in zip(cams, probs):
        ref.add(r.render(cam)[0], p)
    assert np.array_equal(agg.get_raw(), ref.get_raw())
