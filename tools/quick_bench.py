"""Development timing probe (not the judged bench.py): per-stage device times on BASELINE cfg2."""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth  # noqa: E402


def prof(slot):
    ms, n = ctypes.c_double(), ctypes.c_uint64()
    _lib.check(_lib.lib().smesh_profile_read(0, slot, ctypes.byref(ms), ctypes.byref(n)))
    return ms.value, n.value


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 24
    mesh, cams, C = synth.scene(name)
    P = len(mesh.faces)
    W, H = cams[0].resolution
    r = render.triangles(mesh)
    agg = fusion.MeshAggregator(P, C)
    nbuf = 6
    bufs = [synth.device_probs(W, H, C, synth.probs_seed(1, k)) for k in range(nbuf)]
    _lib.synchronize(0)
    for k in range(4):
        agg.fuse_view(r, cams[k], bufs[k % nbuf])
    _lib.synchronize(0)
    for mode in ("plain", "profiled"):
        _lib.check(_lib.lib().smesh_profile_reset(0))
        _lib.check(_lib.lib().smesh_profile_enable(0, 0xFF if mode == "profiled" else 0))
        t0 = time.perf_counter()
        for k in range(views):
            agg.fuse_view(r, cams[k % len(cams)], bufs[k % nbuf])
        _lib.synchronize(0)
        dt = time.perf_counter() - t0
        print("%s %s: %d views in %.3f s -> %.1f views/s (%.3f ms/view)" % (name, mode, views, dt, views / dt, 1e3 * dt / views))
        if mode == "profiled":
            for nm, slot in (("scatter", 0), ("hist", 1), ("raster", 2)):
                ms, n = prof(slot)
                if n:
                    print("  %-8s %6d launches, avg %.1f us" % (nm, n, 1e3 * ms / n))
    _lib.check(_lib.lib().smesh_profile_enable(0, 0))
    # render only
    t0 = time.perf_counter()
    for k in range(views):
        idx, depth = r.render(cams[k % len(cams)])
    _lib.synchronize(0)
    dt = time.perf_counter() - t0
    print("render only: %.3f ms/view" % (1e3 * dt / views))
    t0 = time.perf_counter()
    out = agg.get()
    print("get(): %.1f ms (incl. D2H of %.0f MB); rows touched %d" % (1e3 * (time.perf_counter() - t0), out.nbytes / 1e6, int((out.sum(1) > 0.5).sum())))


if __name__ == "__main__":
    main()


def host_path(name="cfg2", views=10):
    """render() + add(device indices, HOST numpy probs): the reference's usual calling convention
    (colorize_cityscapes_mesh.py:65-67); every view crosses PCIe (pageable memory)."""
    mesh, cams, C = synth.scene(name)
    W, H = cams[0].resolution
    r = render.triangles(mesh)
    agg = fusion.MeshAggregator(len(mesh.faces), C)
    probs = np.asarray(synth.device_probs(W, H, C, 5))
    hwc = np.ascontiguousarray(probs.transpose(1, 0, 2))        # network layout (H,W,C)
    for label, p in (("contiguous (W,H,C)", probs), ("transposed view of (H,W,C)", hwc.transpose(1, 0, 2))):
        idx, _ = r.render(cams[0]); agg.add(idx, p); _lib.synchronize(0)
        t0 = time.perf_counter()
        for k in range(views):
            idx, _ = r.render(cams[k])
            agg.add(idx, p)
        _lib.synchronize(0)
        dt = time.perf_counter() - t0
        print("host probs, %s: %.2f ms/view -> %.1f views/s (%.1f GB/s over PCIe)" % (label, 1e3 * dt / views, views / dt, views * p.nbytes / dt / 1e9))


if __name__ == "__main__" and len(sys.argv) > 3 and sys.argv[3] == "host":
    host_path()
