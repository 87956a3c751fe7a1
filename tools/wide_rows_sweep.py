"""Wide rows (128 <= C <= 1024) on cfg2's mesh and resolution: ms per view of fuse_views (eight views per call), Sum and Summax.
Run with SMESH_WIDE_LIST=0 / 1 to compare k_fuse_tri_wide with k_fuse_tri_wide_list.  usage: python tools/wide_rows_sweep.py [C,C,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
cfg = synth.CONFIGS["cfg2"]; W, H = cfg["width"], cfg["height"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
cams = [synth.ring_camera(k, 40, W, H) for k in range(16)]
Cs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128, 150, 256, 300, 512, 700, 1024]
for C in Cs:
    probs = synth.device_probs(W, H, C, 1, 0.0)
    for kind in ("sum", "summax"):
        agg = fusion.MeshAggregator(len(mesh.faces), C, kind)
        agg.fuse_views(r, cams[:8], [probs] * 8)
        _lib.synchronize(0)
        t0 = time.perf_counter()
        for _ in range(2):
            agg.fuse_views(r, cams[:8], [probs] * 8)
            agg.fuse_views(r, cams[8:], [probs] * 8)
        _lib.synchronize(0)
        dt = (time.perf_counter() - t0) / 32
        print("C=%4d %-7s %-16s %.3f ms/view" % (C, kind, _lib.last_fuse_kernel(), 1e3 * dt), flush=True)
        del agg
    del probs
