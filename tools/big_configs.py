"""Development check of the large BASELINE configs on one GPU: cfg4 (5 M triangles -> texels, 1296x968, C=40)
and cfg5 (20 M triangles, 4096x2160, C=150).  A few views each, timings and sanity properties."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth

def run(name, views, texels=False):
    cfg = synth.CONFIGS[name]
    t0 = time.perf_counter()
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(0, cfg["views"], cfg["views"] // views)][:views]
    print(name, "mesh", len(mesh.faces), "tris built in %.1fs" % (time.perf_counter() - t0), flush=True)
    t0 = time.perf_counter()
    r = render.texels(mesh, cams, 0.1) if texels else render.triangles(mesh)
    P = r.getPrimitivesNum()
    print("  renderer: %d primitives (%.1fs)" % (P, time.perf_counter() - t0), flush=True)
    agg = fusion.MeshAggregator(P, C)
    probs = synth.device_probs(W, H, C, 123, 0.02)
    _lib.synchronize(0)
    for rep in range(2):
        t0 = time.perf_counter()
        for cam in cams:
            agg.fuse_view(r, cam, probs)
        _lib.synchronize(0)
        dt = time.perf_counter() - t0
        print("  pass %d: %.3f ms/view (%s)" % (rep, 1e3 * dt / len(cams), _lib.last_fuse_kernel()), flush=True)
    idx = np.asarray(r.render(cams[0])[0])
    cov = idx != 0xFFFFFFFF
    assert idx[cov].max() < P
    raw = agg.raw_device_array()
    t0 = time.perf_counter()
    out = agg.get()
    touched = out.sum(axis=1) > 0.5
    print("  coverage %.2f, distinct prims view0 %d, touched rows %d, get() %.2fs" % (cov.mean(), len(np.unique(idx[cov])), touched.sum(), time.perf_counter() - t0), flush=True)
    assert np.allclose(out[touched].sum(axis=1), 1.0, rtol=1e-4)

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    if which == "cfg4": run("cfg4", 4, texels=True)
    elif which == "cfg4tri": run("cfg4", 4)
    else: run("cfg5", 3)
