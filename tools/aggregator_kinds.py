"""Development check: fused ms/view on cfg2 for the three aggregators (bench.py times the default, Sum)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
cfg = synth.CONFIGS["cfg2"]; W, H, C = cfg["width"], cfg["height"], cfg["classes"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
probs = synth.device_probs(W, H, C, 1, 0.0)
cams = [synth.ring_camera(k, 50, W, H) for k in range(50)]
for kind in ("sum", "summax", "mul"):
    agg = fusion.MeshAggregator(len(mesh.faces), C, kind)
    for cam in cams[:5]: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for cam in cams: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    print("%-7s %.3f ms/view" % (kind, 1e3 * (time.perf_counter() - t0) / len(cams)), flush=True)
