"""fuse_views (two views per fusion launch) against fuse_view on cfg2: views/s and agreement."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    views = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    if name.startswith("custom:"):      # custom:a,b,W,H,C -- a grid mesh of 2ab triangles seen by the cfg2 camera ring
        ga, gb, cw, ch, C = (int(x) for x in name[7:].split(","))
        mesh = synth.grid_mesh(ga, gb)
        cams = [synth.ring_camera(k, 200, cw, ch) for k in range(200)]
    else:
        mesh, cams, C = synth.scene(name)
    P = len(mesh.faces)
    W, H = cams[0].resolution
    r = render.triangles(mesh)
    nbuf = 8
    bufs = [synth.device_probs(W, H, C, synth.probs_seed(1, k)) for k in range(nbuf)]
    cl = [cams[k % len(cams)] for k in range(views)]
    pl = [bufs[k % nbuf] for k in range(views)]
    res = {}
    for mode in ("single", "batch", "single", "batch"):
        agg = fusion.MeshAggregator(P, C)
        agg.fuse_views(r, cl[:4], pl[:4]) if mode == "batch" else [agg.fuse_view(r, cl[k], pl[k]) for k in range(4)]
        agg.reset()
        _lib.synchronize(0)
        t0 = time.perf_counter()
        if mode == "batch":
            for k in range(0, views, batch):
                agg.fuse_views(r, cl[k:k + batch], pl[k:k + batch])
        else:
            for k in range(views):
                agg.fuse_view(r, cl[k], pl[k])
        _lib.synchronize(0)
        dt = time.perf_counter() - t0
        print("%s %-6s: %d views in %.4f s -> %.1f views/s (%.4f ms/view)" % (name, mode, views, dt, views / dt, 1e3 * dt / views))
        res[mode] = agg.get_raw()
    same = np.array_equal(res["single"].view(np.uint32), res["batch"].view(np.uint32))
    print("raw accumulators bit-equal:", same, " max rel diff %.3g" % float(np.max(np.abs(res["single"] - res["batch"]) / (np.abs(res["single"]) + 1e-20))))


if __name__ == "__main__":
    main()
