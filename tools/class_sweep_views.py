"""ms per view of fuse_views at cfg2's geometry for a class count, with the views of a group in one fusion launch or one launch per view.
usage: python tools/class_sweep_views.py C [C ...]   (run once with SMESH_FUSE_VIEWS=1 for the per-view figure)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from semantic_meshes_amd import _lib, fusion, render, synth
mesh = synth.grid_mesh(1000, 500)
r = render.triangles(mesh)
W, H = 1920, 1080
cams = [synth.ring_camera(k, 16, W, H) for k in range(16)]
for C in [int(x) for x in sys.argv[1:]]:
    agg = fusion.MeshAggregator(len(mesh.faces), C)
    probs = [synth.device_probs(W, H, C, seed=k) for k in range(2)]
    plist = [probs[k % 2] for k in range(16)]
    agg.fuse_views(r, cams[:8], plist[:8]); _lib.synchronize(0)
    t0 = time.perf_counter()
    for rep in range(3):
        agg.fuse_views(r, cams[:8], plist[:8]); agg.fuse_views(r, cams[8:], plist[8:])
    _lib.synchronize(0)
    print("C = %3d: %.4f ms per view (%s, SMESH_FUSE_VIEWS=%s)" % (C, 1e3 * (time.perf_counter() - t0) / 48, _lib.last_fuse_kernel(), os.environ.get("SMESH_FUSE_VIEWS", "default")))
