import sys, os, time
sys.path.insert(0, "/root/repo")
from semantic_meshes_amd import _lib, fusion
for kind in ("sum", "mul"):
  for P, C in ((1_000_000, 19), (1_000_000, 48), (1_000_000, 64), (1_000_000, 128), (500_000, 256), (200_000, 1024), (100_000, 2000), (50_000, 5000)):
    agg = fusion.MeshAggregator(P, C, kind)
    out = agg.get_device(); del out
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for _ in range(5):
        out = agg.get_device(); del out
    _lib.synchronize(0)
    dt = (time.perf_counter() - t0) / 5
    print("%s P = %d, C = %d: get_device %.3f ms -> %.2f TB/s" % (kind, P, C, 1e3 * dt, 8.0 * P * C / dt / 1e12), flush=True)
    del agg
