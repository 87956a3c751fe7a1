"""get() of a cfg2-sized result: where the milliseconds go (kernel 34 us; the rest is the trip to a fresh numpy array)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import semantic_meshes_amd as sm
from semantic_meshes_amd import _lib
P, C = 1000000, 19
agg = sm.fusion.MeshAggregator(P, C)
agg.set_raw(np.random.default_rng(0).random((P, C), dtype=np.float32))
for rep in range(5):
    t = time.perf_counter(); out = agg.get(); dt = time.perf_counter() - t
    print("get() fresh array: %.2f ms" % (1e3 * dt)); del out
t = time.perf_counter(); a = np.empty((P, C), np.float32); a[...] = 0; print("first touch of a fresh 76 MB array: %.2f ms" % (1e3 * (time.perf_counter() - t)))
try:
    print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
except Exception as e:
    print("THP: ?", e)
d = agg.get_device(); _lib.synchronize(0)
t = time.perf_counter(); d = agg.get_device(); _lib.synchronize(0); print("get_device(): %.2f ms" % (1e3 * (time.perf_counter() - t)))
