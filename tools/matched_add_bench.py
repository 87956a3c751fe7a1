"""add() on device-resident COPIES of renders whose originals were exported (sealed): the content match of
smesh_aggregator_add_matched with device-resident probs, ms per cfg2 view (NOTES/round2.md, "Index images that went through
another framework").  Compared with the untouched render (add_rendered) and with matching off (scatter-add)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion, render, synth  # noqa: E402
from semantic_meshes_amd.device import to_device  # noqa: E402


def main():
    mesh, cams, C = synth.scene("cfg2")
    P = len(mesh.faces)
    W, H = cams[0].resolution
    r = render.triangles(mesh)
    agg = fusion.MeshAggregator(P, C)
    bufs = [synth.device_probs(W, H, C, synth.probs_seed(1, k)) for k in range(6)]
    _lib.synchronize(0)
    for label in ("matched", "scatter (matching off)", "untouched render"):
        fusion._MeshAggregator.match_renders = label == "matched"
        best = 1e9
        for rep in range(4):
            planes = [r.render(cams[k])[0] for k in range(6)]
            if label == "untouched render":
                images = planes
            else:
                images = [to_device(np.asarray(p), 0) for p in planes]   # np.asarray exports (seals) the plane
            _lib.synchronize(0)
            t0 = time.perf_counter()
            kernels = set()
            for k, img in enumerate(images):
                agg.add(img, bufs[k])
                kernels.add(_lib.last_fuse_kernel())
            _lib.synchronize(0)
            best = min(best, (time.perf_counter() - t0) / 6)
        print("%-24s %.4f ms per view  %s" % (label, 1e3 * best, sorted(kernels)), flush=True)
    fusion._MeshAggregator.match_renders = True


if __name__ == "__main__":
    main()
