"""Views from INSIDE the scene (the indoor-scan regime of eval-scannet/eval_scannet.py:203-238: triangles from sub-pixel to hundreds of
pixels in one view, some across the near plane): cameras a little above the height field, looking along it.  1080p, C = 19; ms per view
for fuse_views (eight per call), fuse_view, render alone; and the queue lengths of one view.
usage: python tools/close_view_bench.py [a x b meshes, e.g. 1000x500,300x150]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, data, fusion, render, synth   # noqa: E402

W, H, C = 1920, 1080, 19
probs = synth.device_probs(W, H, C, 123, 0.02)
meshes = [(1000, 500), (300, 150)]
if len(sys.argv) > 1:
    meshes = [tuple(int(x) for x in m.split("x")) for m in sys.argv[1].split(",")]
for a, b in meshes:
    mesh = synth.grid_mesh(a, b)
    cams = []
    for k in range(8):
        ang = 2 * np.pi * k / 8
        eye = (3.0 * np.cos(ang), 1.5 * np.sin(ang), 0.6)
        target = (-2.0 * np.cos(ang), -1.0 * np.sin(ang), 0.0)
        R, t = synth.look_at(eye, target, up=(0, 0, 1))
        cams.append(data.Camera(R, t, np.array([W, H]), np.array([0.8 * W, 0.8 * W]), np.array([W / 2.0, H / 2.0])))
    r = render.triangles(mesh)
    agg = fusion.MeshAggregator(len(mesh.faces), C)
    agg.defer = False      # "fuse_view" = one library call per view; the Python layer's default groups such calls by eight (= the fuse_views column)
    out = []
    for name, fn in (("fuse_views", lambda: agg.fuse_views(r, cams, [probs] * 8)),
                     ("fuse_view", lambda: [agg.fuse_view(r, cam, probs) for cam in cams]),
                     ("render", lambda: [r.render(cam, lazy=False) for cam in cams])):
        fn(); fn()
        _lib.synchronize(0)
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        _lib.synchronize(0)
        out.append("%s %.3f" % (name, (time.perf_counter() - t0) / 24 * 1e3))
    r.render(cams[0])
    q = r.render_stats(cams[0])[1]
    idx = np.asarray(r.render(cams[0])[0])
    print("%8d triangles from inside: ms per view: %s   queues of view 0 [boxes over 8 x 8, overflow, huge or clipped, at most 256 pixels] %s, %.0f %% of the pixels covered" % (
        len(mesh.faces), "  ".join(out), q, 100.0 * (idx != 0xFFFFFFFF).mean()), flush=True)
