"""render() + add() per view at cfg2 with the class vectors as a (H,W,C) torch tensor seen as (W,H,C) (permute: a strided device view)
against the dense (W,H,C) image.  usage: python tools/permuted_probs_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from semantic_meshes_amd import _lib, fusion, render, synth
import torch
cfg = synth.CONFIGS["cfg2"]; W, H, C = cfg["width"], cfg["height"], cfg["classes"]
mesh = synth.grid_mesh(cfg["a"], cfg["b"])
r = render.triangles(mesh)
cams = [synth.ring_camera(k, 100, W, H) for k in range(40)]
dense = torch.rand((W, H, C), device="cuda") ; dense = dense / dense.sum(-1, keepdim=True)
hwc = dense.permute(1, 0, 2).contiguous()        # the network's layout
view = hwc.permute(1, 0, 2)                      # (W,H,C) strided
torch.cuda.synchronize()
agg = fusion.MeshAggregator(len(mesh.faces), C)
for label, p in (("dense (W,H,C) device tensor", dense), ("(H,W,C) tensor permuted to (W,H,C), strided", view)):
    for rep in range(2):
        _lib.synchronize(0)
        t0 = time.perf_counter()
        for cam in cams:
            idx, _ = r.render(cam)
            agg.add(idx, p)
        _lib.synchronize(0)
        dt = (time.perf_counter() - t0) / len(cams)
    print("render + add, %-48s %.3f ms/view (%s, %s)" % (label, 1e3 * dt, _lib.last_add_path(), _lib.last_fuse_kernel()), flush=True)
