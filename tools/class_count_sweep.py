"""Development check: fused ms/view on the cfg2 mesh / resolution for different class counts (which kernel, how fast)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
cfg = synth.CONFIGS["cfg2"]; W, H = cfg["width"], cfg["height"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
cams = [synth.ring_camera(k, 40, W, H) for k in range(40)]
for C in (5, 13, 19, 21, 27, 32, 40, 41, 48, 49, 64, 100, 127, 150, 256):
    probs = synth.device_probs(W, H, C, 1, 0.0)
    agg = fusion.MeshAggregator(len(mesh.faces), C)
    for cam in cams[:4]: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for cam in cams: agg.fuse_view(r, cam, probs)
    _lib.synchronize(0)
    dt = (time.perf_counter() - t0) / len(cams)
    nbytes = 4.0 * W * H * C * 0.64 + 8.0 * C * 0.64e6   # visible probs rows + accumulator rows of ~0.64 M visible primitives
    print("C=%4d  %-16s %.3f ms/view  (~%.0f MB of rows -> %.2f TB/s incl. the rasteriser's 0.05 ms)" % (
        C, _lib.last_fuse_kernel(), 1e3 * dt, nbytes / 1e6, nbytes / dt / 1e12), flush=True)
    del probs, agg
