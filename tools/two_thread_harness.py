"""The reference harness's loop (python/scripts/eval_scannet.py:189-238): the main thread renders view k + 1 while a worker thread
adds view k.  Times, at cfg2's geometry (1 M triangles, 1080p, 19 classes), `views` views:
  serial        idx = render(cam); add(idx, probs)                          one thread
  two threads   main: render -> queue(maxsize 2); worker: add               the harness
  fuse_view     the fused entry point, for reference
with the class vectors resident in HBM (`device`) and as host numpy arrays (`host`: what the harness has -- the network's output).
Run on the GPU box:  python tools/two_thread_harness.py [views]"""
import os
import queue
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import semantic_meshes_amd as sm          # noqa: E402
from semantic_meshes_amd import _lib, synth   # noqa: E402


def main():
    views = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = synth.CONFIGS["cfg2"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    cams = [synth.ring_camera(k, 200, W, H) for k in range(views)]
    r = sm.render.triangles(mesh)
    P = len(mesh.faces)
    dev = [synth.device_probs(W, H, C, synth.probs_seed(1, k), 0.0, 0) for k in range(8)]
    host = [np.asarray(p).copy() for p in dev[:4]]

    def serial(agg, probs):
        for k, cam in enumerate(cams):
            idx, _ = r.render(cam)
            agg.add(idx, probs[k % len(probs)])

    def threaded(agg, probs):
        q = queue.Queue(maxsize=2)

        def worker():
            while True:
                item = q.get()
                if item is None:
                    return
                agg.add(*item)

        t = threading.Thread(target=worker)
        t.start()
        for k, cam in enumerate(cams):
            idx, _ = r.render(cam)
            q.put((idx, probs[k % len(probs)]))
        q.put(None)
        t.join()

    def fused(agg, probs):
        for k, cam in enumerate(cams):
            agg.fuse_view(r, cam, probs[k % len(probs)])

    results = {}
    # `device`: the default -- render() planes rasterised on first use, views with the library's own device arrays deferred into groups of
    # eight (fusion.py); `device/call`: the same loops with MeshAggregator.defer = False (every add() / fuse_view() is its own library
    # call, as until round 5)
    for where, probs, defer in (("device", dev, True), ("device/call", dev, False), ("host", host, True)):
        for name, fn in (("serial", serial), ("two threads", threaded), ("fuse_view", fused)):
            best, raws = None, []
            for rep in range(4):
                agg = sm.fusion.MeshAggregator(P, C)
                agg.defer = defer
                _lib.synchronize(0)
                t0 = time.perf_counter()
                fn(agg, probs)
                _lib.synchronize(0)
                dt = time.perf_counter() - t0
                if rep:
                    best = dt if best is None else min(best, dt)
                raws.append(agg.get_raw())
            same = all(np.array_equal(raws[0], x) for x in raws[1:])
            results[(where, name)] = (best, raws[0])
            print("%-11s %-12s %8.3f ms per view  (%7.0f views/s)  repeats bit-equal: %s" % (where, name, best / views * 1e3, views / best, same), flush=True)
        a, b = results[(where, "serial")][1], results[(where, "two threads")][1]
        print("%-11s two threads == serial bit for bit: %s" % (where, np.array_equal(a, b)), flush=True)
    print("device (deferred groups) == device/call (one library call per view) bit for bit: %s"
          % np.array_equal(results[("device", "serial")][1], results[("device/call", "serial")][1]), flush=True)


if __name__ == "__main__":
    main()
