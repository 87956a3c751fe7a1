"""Development check: how much does the triangle ORDER of the mesh matter?  cfg2 mesh with faces in grid order, shuffled inside
windows of 4096 faces, and fully shuffled."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth, data
cfg = synth.CONFIGS["cfg2"]
W, H, C = cfg["width"], cfg["height"], cfg["classes"]
base = synth.grid_mesh(cfg["a"], cfg["b"])
probs = synth.device_probs(W, H, C, 123, 0.0)
cams = [synth.ring_camera(k, 8, W, H) for k in range(8)]
rng = np.random.default_rng(0)
F = len(base.faces)
orders = {"grid order": np.arange(F)}
win = np.arange(F).reshape(-1, 4000).copy()
for row in win: rng.shuffle(row)
orders["shuffled in windows of 4000"] = win.reshape(-1)
orders["fully shuffled"] = rng.permutation(F)
vperm = rng.permutation(len(base.vertices))          # new position of every vertex
vinv = np.empty_like(vperm); vinv[vperm] = np.arange(len(vperm))
meshes = {name: data.Mesh(base.vertices, base.faces[perm]) for name, perm in orders.items()}
meshes["grid faces, shuffled VERTEX numbering"] = data.Mesh(base.vertices[vinv], vperm[base.faces].astype(np.int32))
for name, mesh in meshes.items():
    r = render.triangles(mesh); agg = fusion.MeshAggregator(F, C)
    for cam in cams[:2]: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for _ in range(3):
        for cam in cams: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    print("%-30s %.3f ms/view" % (name, 1e3 * (time.perf_counter() - t0) / 24), flush=True)
