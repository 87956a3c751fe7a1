"""Scratch: wall time of consecutive plain render() calls (after some fuse_view calls)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
name, texels = sys.argv[1], len(sys.argv) > 2 and sys.argv[2] == "texels"
cfg = synth.CONFIGS[name]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
W, H, C = cfg["width"], cfg["height"], cfg["classes"]
cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(0, cfg["views"], cfg["views"] // 4)][:4]
r = render.texels(mesh, cams, 0.1) if texels else render.triangles(mesh)
agg = fusion.MeshAggregator(r.getPrimitivesNum(), C)
probs = synth.device_probs(W, H, C, 123, 0.02)
for rep in range(2):
    for cam in cams:
        agg.fuse_view(r, cam, probs)
_lib.synchronize(0)
for i in range(6):
    t0 = time.perf_counter()
    idx, depth = r.render(cams[i % 4])
    _lib.synchronize(0)
    print("render %d: %.3f ms" % (i, 1e3 * (time.perf_counter() - t0)), flush=True)
    del idx, depth
for cam in cams:
    t0 = time.perf_counter(); agg.fuse_view(r, cam, probs); _lib.synchronize(0)
    print("fuse_view: %.3f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
