"""One-off: the randomised differential tests of tests/test_gpu_fuzz.py over seeds beyond the committed ones.
usage (GPU box): python tools/soup_sweep.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import semantic_meshes_amd as sm
from oracle import oracle
import test_gpu_fuzz as fz
oracle.lib(); oracle.set_threads(1); oracle.set_accum_double(False)
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 300)
only = os.environ.get("SMESH_SWEEP_ONLY")            # e.g. "room": only the generators whose name contains it
tests = [getattr(fz, n) for n in dir(fz) if n.startswith("test_random") and "forced_reorder" not in n and (only in n if only else "dense" not in n)]   # (that one is a subprocess wrapper)
fails, t0 = 0, time.time()
for seed in range(first, first + count):
    for t in tests:
        try:
            t(sm, oracle, seed)
        except Exception as e:
            fails += 1
            oracle.set_accum_double(False)
            print("FAIL %s seed %d: %s" % (t.__name__, seed, str(e).splitlines()[0][:200]), flush=True)
print("%d seeds x %d generators (%s), %d failures, %.0f s" % (count, len(tests), ", ".join(t.__name__ for t in tests), fails, time.time() - t0), flush=True)
