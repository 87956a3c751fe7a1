"""add() on foreign device images at cfg2's geometry over class counts: image records + triangle-order fusion against the atomic
scatter-add.  usage: python tools/generic_add_sweep.py  (run once per path: SMESH_ADD_RECORDS=0 for the scatter-add)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from semantic_meshes_amd import _lib, fusion, render, synth
from semantic_meshes_amd.device import to_device
cfg = synth.CONFIGS["cfg2"]; W, H = cfg["width"], cfg["height"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
P = r.getPrimitivesNum()
cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(12)]
images = [to_device(np.asarray(r.render(cam)[0])) for cam in cams]
del r
fusion._MeshAggregator.match_renders = False
out = []
for C in [int(c) for c in (sys.argv[1:] or "5 13 19 27 32 40 48 64 100 150".split())]:
    probs = synth.device_probs(W, H, C, 1, 0.0)
    agg = fusion.MeshAggregator(P, C)
    for rep in range(3):
        _lib.synchronize(0)
        t0 = time.perf_counter()
        for img in images:
            agg.add(img, probs)
        _lib.synchronize(0)
        dt = (time.perf_counter() - t0) / len(images)
    out.append("C=%d %.3f" % (C, 1e3 * dt))
    del agg, probs
print("%s (%s): ms/view %s" % (_lib.last_add_path(), _lib.last_fuse_kernel(), "  ".join(out)), flush=True)
