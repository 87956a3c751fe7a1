"""fuse_views on a mesh of medium triangles: 300 x 150 quads = 90 000 triangles at 1080p (~23 pixels per triangle, boxes just over
8 x 8) -- a decimated indoor scan's regime.  usage: python tools/medium_mesh_bench.py [a b]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 150)
W, H, C = 1920, 1080, 19
probs = synth.device_probs(W, H, C, 123, 0.02)
mesh = synth.grid_mesh(a, b)
cams = [synth.ring_camera(k, 8, W, H) for k in range(8)]
r = render.triangles(mesh)
agg = fusion.MeshAggregator(len(mesh.faces), C)
for _ in range(2):
    agg.fuse_views(r, cams, [probs] * 8)
_lib.synchronize(0)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    agg.fuse_views(r, cams, [probs] * 8)
_lib.synchronize(0)
print("%d triangles: fuse_views %.3f ms/view (%s)" % (len(mesh.faces), 1e3 * (time.perf_counter() - t0) / (8 * reps), _lib.lib().smesh_last_fuse_kernel().decode()))
