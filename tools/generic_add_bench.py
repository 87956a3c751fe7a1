"""add() on index images the library did not render (device-resident copies of renders, so that no render matches by identity;
content matching off): image records + triangle-order fusion against the atomic scatter-add (SMESH_ADD_RECORDS=0).
usage: python tools/generic_add_bench.py [cfg2|cfg4|cfg5] [views]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from semantic_meshes_amd import _lib, fusion, render, synth
from semantic_meshes_amd.device import to_device
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 16
cfg = synth.CONFIGS[name]; W, H, C = cfg["width"], cfg["height"], cfg["classes"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.texels(mesh, [synth.ring_camera(k, cfg["views"], W, H) for k in range(cfg["views"])], 0.1) if cfg.get("texels") else render.triangles(mesh)
P = r.getPrimitivesNum()
probs = synth.device_probs(W, H, C, 1, 0.0)
cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(nv)]
images = [to_device(np.asarray(r.render(cam)[0])) for cam in cams]      # copies: not the renderer's planes
del r
fusion._MeshAggregator.match_renders = False
agg = fusion.MeshAggregator(P, C)
for rep in range(3):
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for img in images:
        agg.add(img, probs)
    _lib.synchronize(0)
    dt = (time.perf_counter() - t0) / len(images)
N = W * H
touched = int((agg.get_raw().sum(1) != 0).sum())
bytes_per_view = 4 * N + 4 * N * C + 8 * C * touched
print("%s: add() on foreign device images: %.3f ms/view, path %s, kernel %s, 8(d) bytes %.1f MB -> %.2f TB/s = %.2f of 8 TB/s" % (
    name, 1e3 * dt, _lib.last_add_path(), _lib.last_fuse_kernel(), bytes_per_view / 1e6,
    bytes_per_view / dt / 1e12, bytes_per_view / dt / 8e12), flush=True)
