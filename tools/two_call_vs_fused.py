"""Development check: the reference's two-call loop (render, then add) against fuse_view, cfg2, device-resident probs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
cfg = synth.CONFIGS["cfg2"]; W, H, C = cfg["width"], cfg["height"], cfg["classes"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
probs = synth.device_probs(W, H, C, 1, 0.0)
cams = [synth.ring_camera(k, 100, W, H) for k in range(100)]
agg = fusion.MeshAggregator(len(mesh.faces), C)
for mode in ("fuse_view", "render + add", "render + add (index image exported first: recognised by content)", "render + add (exported, SMESH_MATCH off: generic scatter-add)"):
    for rep in range(2):
        _lib.synchronize(0)
        t0 = time.perf_counter()
        for cam in cams:
            if mode == "fuse_view":
                agg.fuse_view(r, cam, probs)
            else:
                idx, depth = r.render(cam)
                if "exported" in mode:
                    _ = idx.__cuda_array_interface__
                fusion._MeshAggregator.match_renders = "MATCH off" not in mode
                agg.add(idx, probs)
        _lib.synchronize(0)
        dt = (time.perf_counter() - t0) / len(cams)
    print("%-70s %.3f ms/view (%s)" % (mode, 1e3 * dt, _lib.last_fuse_kernel()), flush=True)
