"""The address stream of the triangle-order fusion, taken from real renders (round 6): for `views` views of a BASELINE
workload, the pixels of every triangle in triangle order -- exactly the class-vector rows wave w of k_fuse_tri_wide (triangles 64 w .. 64 w + 63)
reads, in the order it reads them.  `tools/stream_bench p <file>` replays it with nothing else around it.
usage: python tools/dump_pixel_stream.py <workload> <views> <out file>"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import render, synth   # noqa: E402

workload, views, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = synth.CONFIGS[workload]
W, H = cfg["width"], cfg["height"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
F = len(mesh.faces)
waves = (F + 63) // 64
r = render.triangles(mesh)
merged_tri, merged_ent = [], []
with open(out, "wb") as f:
    np.array([0x534D5253, views, W * H, waves], np.uint32).tofile(f)
    for k in range(views):
        cam = synth.ring_camera(4 + k, cfg["views"], W, H)       # (the bench's first timed group)
        idx = np.asarray(r.render(cam, lazy=False)[0]).ravel()   # pixel p = x * H + y holds the triangle it shows
        order = np.argsort(idx, kind="stable").astype(np.uint32) # pixels by triangle, a triangle's pixels in image order (x, then y)
        tri = idx[order]
        n = int(np.searchsorted(tri, F))                         # (background 0xFFFFFFFF sorts last)
        start = np.searchsorted(tri[:n], np.arange(0, waves + 1, dtype=np.uint64) * 64).astype(np.uint32)
        start.tofile(f)
        order[:n].tofile(f)
        merged_tri.append(tri[:n].astype(np.uint32))
        merged_ent.append(order[:n] | np.uint32(k << 29))
        print("view %d: %d visible pixels of %d, %d of %d triangles seen" % (k, n, W * H, len(np.unique(tri[:n])), F), flush=True)
    # ... and the same rows in the order the KERNEL walks them: triangle after triangle, a triangle's views in order (stable sort by triangle of
    # the views' lists laid end to end) -- uint32 start[waves + 1], uint32 entry[n] (pixel | view << 29), uint32 triangle[n]
    tri_all = np.concatenate(merged_tri)
    ent_all = np.concatenate(merged_ent)
    o = np.argsort(tri_all, kind="stable")
    tri_all, ent_all = tri_all[o], ent_all[o]
    start = np.searchsorted(tri_all, np.arange(0, waves + 1, dtype=np.uint64) * 64).astype(np.uint32)
    start.tofile(f)
    ent_all.tofile(f)
    tri_all.tofile(f)
    print("merged: %d rows of %d triangles" % (len(tri_all), len(np.unique(tri_all))), flush=True)
