"""Hunt for the intermittent wrong result of the sharded path (NOTES/round4.md "One unexplained failure", NOTES/round5.md 1).

Not collected by pytest (no test_ prefix): run on the GPU box as

    python tests/flake_hunt.py fuse --workers 8 --iters 2000 --hog 1      # the fusion alone, under contention
    python tests/flake_hunt.py exchange --repeats 60 --hog 1              # the test's own worker (gloo exchange), looped

`fuse`: `workers` processes share the GPU (plus `hog` processes that keep every CU busy with cfg2-sized fusion launches).  Each
worker owns view `rank` of the failing test's scene (3 600 triangles at 320 x 240, ~21 pixels each: the medium-triangle regime) and
repeats, with fresh aggregators and fresh class-vector images every time:
  single   fuse_views of its ONE view (k_fuse_tri<19, ., 1 view> with the medium-triangle waves beside the main waves) -> get_raw()
  rows     get_rows(lo, hi) of that aggregator against the rows of get_raw() normalised on the host
  all8     fuse_views of all eight views -> get_raw()
  ranged   fuse_views_ranged(nparts = 3) of all eight views -> get_raw()
Every result is compared with the first iteration's (itself checked against the float64 oracle): the float atomics of the medium
triangles have no fixed order, so the bar is 2e-6 of the row's largest element -- a lost or doubled contribution is 1e-1.
A mismatch is dumped to gpurun_out/flake/ and counted; the run goes on.  Exit status 1 if anything was found.
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "gpurun_out", "flake")


def hog(seconds):
    """Keeps the GPU busy with the headline workload (1 M triangles, 1080p, 19 classes, eight views per call) for `seconds`."""
    import semantic_meshes_amd as sm
    from semantic_meshes_amd import synth
    cfg = synth.CONFIGS["cfg2"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    cams = [synth.ring_camera(k, 200, W, H) for k in range(8)]
    renderer = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(len(mesh.faces), C)
    probs = [synth.device_probs(W, H, C, synth.probs_seed(1, k), 0.0, 0) for k in range(8)]
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            agg.fuse_views(renderer, cams, probs)
        agg.get_rows(0, 64)      # (a host synchronisation now and then: the queue stays bounded)
        n += 160
    print("hog: %d cfg2 views fused in %.0f s" % (n, time.time() - t0), flush=True)


def normalise(raw):
    s = np.abs(raw).sum(axis=1, keepdims=True, dtype=np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        out = raw / s
    out[~np.isfinite(out)] = 0.0
    return out


def worker(rank, world, iters, check_oracle):
    import semantic_meshes_amd as sm
    from semantic_meshes_amd import distributed as smdist, synth
    from helpers import small_scene
    mesh, cams = small_scene(60, 30, 320, 240, views=8)
    P, C = len(mesh.faces), 19
    renderer = sm.render.triangles(mesh)
    k = rank % len(cams)
    lo, hi = smdist.owned_rows(P, rank % 8, 8)

    # bisection switches: HUNT_POOL=1 -- the class-vector images are written into buffers that are allocated once and reused (no
    # hipMalloc / hipFree per image); HUNT_SYNC=1 -- the device is idle when the generator kernel is launched into a fresh allocation
    pool_on, sync_on = os.environ.get("HUNT_POOL") == "1", os.environ.get("HUNT_SYNC") == "1"
    pool = {}

    def probs_of_view(v, slot=0):
        W, H = cams[v].resolution
        if pool_on:
            if (v, slot) not in pool:
                pool[(v, slot)] = synth.device_probs(W, H, C, synth.probs_seed(3, v), 0.05, 0)
            from semantic_meshes_amd.device import DeviceArray
            base = pool[(v, slot)]
            _lib_fill(base)                       # (overwritten first, so that a generator kernel that did not write shows)
            return synth.device_probs(W, H, C, synth.probs_seed(3, v), 0.05, 0, out=base)
        if sync_on:
            from semantic_meshes_amd import _lib
            from semantic_meshes_amd.device import DeviceBuffer
            buf = DeviceBuffer(W * H * C * 4, 0)
            _lib.synchronize(0)
            return synth.device_probs(W, H, C, synth.probs_seed(3, v), 0.05, 0, out=buf.view((W, H, C), np.float32))
        return synth.device_probs(W, H, C, synth.probs_seed(3, v), 0.05, 0)

    def _lib_fill(arr):
        # all-zero rows are what the failure looks like: a reused buffer is cleared before the generator runs
        import ctypes
        from semantic_meshes_amd import _lib
        z = np.zeros(arr.shape, np.float32)
        _lib.check(_lib.lib().smesh_memcpy(ctypes.c_void_p(arr.ptr), z.ctypes.data_as(ctypes.c_void_p), z.nbytes, _lib.MEM_DEVICE, _lib.MEM_HOST, 0))

    host_probs = {}        # view -> host copy of its class vectors as the generator wrote them the first time (checked against the oracle's)
    kept = {}              # the device images of the current iteration's `all8` leg, alive until the leg has been compared

    def legs(kind):
        out = {}
        a = sm.fusion.MeshAggregator(P, C, kind)
        a.fuse_views(renderer, [cams[k]], [probs_of_view(k)])
        out["single"] = a.get_raw()
        out["rows"] = a.get_rows(lo, hi)
        b = sm.fusion.MeshAggregator(P, C, kind)
        kept["all8"] = [probs_of_view(v, 1) for v in range(len(cams))]
        b.fuse_views(renderer, cams, kept["all8"])
        out["all8"] = b.get_raw()
        if not host_probs:
            for v, p in enumerate(kept["all8"]):
                host_probs[v] = np.asarray(p).copy()
        c = sm.fusion.MeshAggregator(P, C, kind)
        c.fuse_views_ranged(renderer, cams, [probs_of_view(v, 2) for v in range(len(cams))], nparts=3)
        out["ranged"] = c.get_raw()
        return out

    ref = {kind: legs(kind) for kind in ("sum", "summax")}
    if check_oracle:
        from oracle import oracle
        oracle.set_threads(1)
        oracle.set_accum_double(True)
        o_r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
        for kind in ("sum", "summax"):
            one, all8 = oracle.OracleAggregator(P, C, kind), oracle.OracleAggregator(P, C, kind)
            for v, cam in enumerate(cams):
                W, H = cam.resolution
                idx = o_r.render(cam)[0]
                pr = oracle.synth_probs(W * H, C, synth.probs_seed(3, v), 0.05).reshape(W, H, C)
                all8.add(idx, pr)
                assert np.array_equal(host_probs[v], pr), "the generator's image differs from the oracle's"
                if v == k:
                    one.add(idx, pr)
            for name, want in (("single", one.get_raw()), ("all8", all8.get_raw()), ("ranged", all8.get_raw())):
                got = ref[kind][name]
                assert np.all(np.abs(got - want) <= 1e-5 * np.abs(want).max(axis=1, keepdims=True) + 1e-7), (rank, kind, name, "reference iteration differs from the oracle")
    bad = 0
    t0 = time.time()
    for it in range(iters):
        for kind in ("sum", "summax"):
            got = legs(kind)
            for name, g in got.items():
                want = normalise(got["single"])[lo:hi] if name == "rows" else ref[kind][name]
                scale = np.abs(want).max(axis=1, keepdims=True)
                wrong = (np.abs(g - want) > 2e-6 * scale + 1e-9).any(axis=1)
                if wrong.any():
                    bad += 1
                    path = os.path.join(OUT, "rank%d_%s_%s_it%d.npz" % (rank, kind, name, it))
                    np.savez(path, got=g, want=want, rows=np.nonzero(wrong)[0], single=got["single"])
                    print("FLAKE rank %d iteration %d kind %s leg %s: %d rows wrong (first %s), worst ratio %.3g -> %s" % (
                        rank, it, kind, name, wrong.sum(), np.nonzero(wrong)[0][:8].tolist(),
                        float((np.abs(g - want) / np.maximum(scale, 1e-30)).max()), path), flush=True)
                    if name == "all8":      # were the INPUTS what the generator should have written?
                        for v, p in enumerate(kept["all8"]):
                            now = np.asarray(p)
                            differ = (now != host_probs[v]).any(axis=2)
                            if differ.any():
                                flat = np.nonzero(differ.reshape(-1))[0]
                                zero = (now.reshape(-1, C)[flat] == 0).all(axis=1).sum()
                                print("   view %d: %d pixels of the class-vector image differ from the generator's output (%d of them all zero); "
                                      "pixel offsets %d .. %d, byte offsets %d .. %d of the allocation at 0x%x; 256-pixel chunks (one workgroup of the generator each) "
                                      "%d, their numbers mod 8: %s" % (
                                          v, flat.size, zero, flat[0], flat[-1], flat[0] * C * 4, flat[-1] * C * 4 + C * 4, p.ptr,
                                          np.unique(flat // 256).size, np.unique((flat // 256) % 8).tolist()), flush=True)
                            else:
                                print("   view %d: class vectors intact when read back" % v, flush=True)
    print("worker %d: %d iterations, %d mismatches, %.0f s" % (rank, iters, bad, time.time() - t0), flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["fuse", "exchange", "_worker", "_hog"])
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--iters", type=int, default=500)
    ap.add_argument("--repeats", type=int, default=40)
    ap.add_argument("--hog", type=int, default=1)
    ap.add_argument("--hog-seconds", type=float, default=3600.0)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--timeout", type=int, default=1500)
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    if args.mode == "_hog":
        hog(args.hog_seconds)
        return 0
    if args.mode == "_worker":
        return 1 if worker(args.rank, args.workers, args.iters, check_oracle=(args.rank == 0)) else 0
    hogs = [subprocess.Popen([sys.executable, __file__, "_hog", "--hog-seconds", str(args.hog_seconds)]) for _ in range(args.hog)]
    status = 0
    try:
        if args.mode == "fuse":
            procs = [subprocess.Popen([sys.executable, __file__, "_worker", "--rank", str(r), "--workers", str(args.workers), "--iters", str(args.iters)])
                     for r in range(args.workers)]
            for p in procs:
                try:
                    status |= 1 if p.wait(timeout=args.timeout) else 0
                except subprocess.TimeoutExpired:
                    p.kill()
                    print("worker timed out", flush=True)
                    status |= 2
        else:
            import tempfile
            import test_gpu_sharded as tgs
            with tempfile.TemporaryDirectory() as tmp:
                codes, outs, dump_dir = tgs.run_sharded_workers(tmp, 8, repeats=args.repeats, timeout=args.timeout)
            for r, (c, o) in enumerate(zip(codes, outs)):
                print("rank %d exit %s: %s" % (r, c, o[-600:].replace("\n", " | ")), flush=True)
            if any(codes):
                status = 1
                print("\n".join(tgs.diagnose_sharded_dumps(dump_dir)), flush=True)
    finally:
        for h in hogs:
            h.kill()
    print("flake_hunt %s: %s" % (args.mode, "CLEAN" if status == 0 else "FOUND SOMETHING (status %d)" % status), flush=True)
    return status


if __name__ == "__main__":
    sys.exit(main())
