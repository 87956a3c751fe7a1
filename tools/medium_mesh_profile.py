"""Where a group of eight views of a medium mesh goes: fuse_views on a grid mesh of a x b quads at 1080p, C = 19, twenty groups (run under
rocprofv3 --kernel-trace --stats).  usage: python tools/medium_mesh_profile.py [a b]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (300, 150)
single = len(sys.argv) > 3 and sys.argv[3] == "single"      # one view per call: render() + add(), the reference's literal loop
W, H, C = 1920, 1080, 19
probs = synth.device_probs(W, H, C, 123, 0.02)
mesh = synth.grid_mesh(a, b)
cams = [synth.ring_camera(k, 8, W, H) for k in range(8)]
r = render.triangles(mesh)
agg = fusion.MeshAggregator(len(mesh.faces), C)
for _ in range(20):
    if single:
        for cam in cams:
            idx, _depth = r.render(cam)
            agg.add(idx, probs)
    else:
        agg.fuse_views(r, cams, [probs] * 8)
_lib.synchronize(0)
