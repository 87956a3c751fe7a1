"""Development check: fused views/s at 1080p, C = 19, for grid meshes of different density (triangle size on screen).
Small triangles (box <= 8 x 8) take the lane-per-triangle paths, larger ones the cooperative big-triangle paths."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth

W, H, C = 1920, 1080, 19
probs = synth.device_probs(W, H, C, 123, 0.02)
meshes = [(1000, 500), (700, 350), (500, 250), (300, 150), (200, 100), (100, 50), (30, 15)]
if len(sys.argv) > 1:      # e.g. 200x100,300x150
    meshes = [tuple(int(x) for x in m.split("x")) for m in sys.argv[1].split(",")]
for a, b in meshes:
    mesh = synth.grid_mesh(a, b)
    cams = [synth.ring_camera(k, 8, W, H) for k in range(8)]
    r = render.triangles(mesh)
    agg = fusion.MeshAggregator(len(mesh.faces), C)
    agg.defer = False      # "fuse_view" below = one library call per view (the Python layer would group them by eight: the fuse_views column)
    for cam in cams[:2]:
        agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        for cam in cams:
            agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    dt = (time.perf_counter() - t0) / (reps * len(cams))
    for cam in cams[:2]:
        agg.fuse_views(r, cams, [probs] * len(cams))
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        agg.fuse_views(r, cams, [probs] * len(cams))
    _lib.synchronize(0)
    dv = (time.perf_counter() - t0) / (reps * len(cams))
    t0 = time.perf_counter()
    for cam in cams:
        r.render(cam, lazy=False)
    _lib.synchronize(0)
    dr = (time.perf_counter() - t0) / len(cams)
    print("%8d triangles: fuse_view %8.3f ms/view   fuse_views (8 per call) %8.3f ms/view   (render alone %7.3f ms)" % (len(mesh.faces), 1e3 * dt, 1e3 * dv, 1e3 * dr), flush=True)
