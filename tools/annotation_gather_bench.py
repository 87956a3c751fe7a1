"""ModelRenderer::render (Mesh.h:25-42) on the GPU: gather of the fused annotation rows back to a (W,H,C) image, device in / device out.
usage: python tools/annotation_gather_bench.py [cfg2|cfg5]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from semantic_meshes_amd import _lib, fusion, render, synth
from semantic_meshes_amd.device import DeviceBuffer
from semantic_meshes_amd.fusion import _c64
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = synth.CONFIGS[name]; W, H, C = cfg["width"], cfg["height"], cfg["classes"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
P = r.getPrimitivesNum()
agg = fusion.MeshAggregator(P, C)
cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(4)]
probs = synth.device_probs(W, H, C, 1, 0.0)
for cam in cams:
    agg.fuse_view(r, cam, probs)
mr = fusion.ModelRenderer(agg)
idx, _ = r.render(cams[1])
out = DeviceBuffer(W * H * C * 4, 0)
bg = np.zeros(C, np.float32)
def once():
    _lib.check(_lib.lib().smesh_annotation_renderer_render(mr._h, ctypes.c_void_p(idx.ptr), 0, _c64((H, 1)), _lib.MEM_DEVICE,
               bg.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(out.ptr), _lib.MEM_DEVICE, W, H))
once()
t0 = time.perf_counter()
for _ in range(20):
    once()
dt = (time.perf_counter() - t0) / 20
nbytes = 4 * W * H + 2 * 4 * W * H * C
print("%s: annotation gather %.3f ms per image (host-timed, synchronous call), %.1f MB read+written -> %.2f TB/s" % (name, 1e3 * dt, nbytes / 1e6, nbytes / dt / 1e12))
