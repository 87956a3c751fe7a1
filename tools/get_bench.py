"""get() on the device (normalised float32[P,C] left in HBM): k_finalize_tile.  usage: python tools/get_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_meshes_amd import _lib, fusion
for P, C in ((1_000_000, 19), (5_000_000, 40), (20_000_000, 150), (2_000_000, 300)):
    agg = fusion.MeshAggregator(P, C)
    out = agg.get_device(); del out
    _lib.synchronize(0)
    t0 = time.perf_counter()
    for _ in range(5):
        out = agg.get_device(); del out
    _lib.synchronize(0)
    dt = (time.perf_counter() - t0) / 5
    print("P = %d, C = %d: get_device %.3f ms (%.1f MB read + written -> %.2f TB/s, allocation included)" % (P, C, 1e3 * dt, 8e-6 * P * C, 8.0 * P * C / dt / 1e12), flush=True)
    del agg
