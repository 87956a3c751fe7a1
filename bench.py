#!/usr/bin/env python3
"""bench.py -- views/sec fused on the BASELINE.json headline workload (cfg2: 1 M-triangle mesh, 1920x1080,
19 classes, 200 views per GPU), plus the roofline of the dominant kernel and a CPU baseline.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched through
`python -m torch.distributed.run`, one rank per GPU.  A "step" is one pass of the hot path over one view:
render(camera) -> add(indices, probs), with the view's class-probability image already resident in HBM (generated on
the device before the timed region).  The views are handed to the library `--views-per-call` at a time
(smesh_fuse_views: same results as one smesh_fuse_view call per view, but the views of a group share their kernel
launches; `--views-per-call 1` makes one smesh_fuse_view call per view).  After the K steps of every rank the raw
accumulators are summed with ONE all-reduce (RCCL), inside the timed region: the native `smesh_allreduce`, enqueued on the
library's own stream right behind the last fusion kernel (`SMESH_ALLREDUCE=torch` selects the torch.distributed plumbing
instead).  The timed region contains exactly one host synchronisation: the closing barrier.  Rank 0 prints ONE JSON line.

`--workload cfg4` (5 M triangles as texel primitives, 1296x968, 40 classes) and `--workload cfg5` (20 M triangles,
4096x2160, 150 classes) run the other single-GPU BASELINE configs through the same code and print the same JSON.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from semantic_meshes_amd import _lib, fusion, render, synth  # noqa: E402
from semantic_meshes_amd import comm as smcomm  # noqa: E402
from semantic_meshes_amd import distributed as smdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)

KERNEL_NOTES = {
    "k_fuse_tri": "k_fuse_tri (triangle-order fusion: gather + accumulate, one owner per accumulator row)",
    "k_fuse_tri_any": "k_fuse_tri_any (triangle-order fusion, row split over lane groups)",
    "k_fuse_tri_wide": "k_fuse_tri_wide (triangle-order fusion, a row per wave: 128 <= C <= 1024)",
    "k_fuse_texel": "k_fuse_texel (triangle-order fusion of texel primitives)",
    "k_scatter_strip": "k_scatter_strip (segmented scatter-add)",
}


def prof_read(device, slot):
    """(total ms, regions timed, dominant-kernel launches inside them, views those launches fused)"""
    ms, r, n, v = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
    _lib.check(_lib.lib().smesh_profile_read_ex(device, slot, ctypes.byref(ms), ctypes.byref(r), ctypes.byref(n), ctypes.byref(v)))
    return ms.value, int(r.value), int(n.value), int(v.value)


def pmc_traffic(workload, kernel_prefix, steps, warmup):
    """HBM bytes per launch of every instance of the dominant kernel, measured NOW (`--pmc`): two short runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes (the TCC
    block cannot count both at once) -- and traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950: FETCH_SIZE counts 64 B per
    128-byte request), averaged over the launches of each kernel instance.  The passes run `--steps steps --warmup warmup` with the group
    pipeline off (one kernel at a time on the GPU), so that they launch the same instances (8 / 4 / 2 / 1 views) as the caller's run.
    Returns ({views per launch (0: not a template instance) -> bytes per launch}, note)."""
    import csv
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found"
    per = {}
    tmp = tempfile.mkdtemp(prefix="smesh_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "bench", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", str(steps), "--warmup", str(warmup), "--repeats", "1",
                   "--no-cpu-baseline", "--no-host-path", "--no-pmc", "--no-group-pipeline"]
            env = dict(os.environ, TMPDIR="/tmp", SMESH_BENCH_PROBS_GB="40")
            env.pop("RANK", None)
            try:
                subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=900 if workload in ("cfg5", "cfg4t") else 300)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s timed out" % counter
            path = None
            for dirpath, _, files in os.walk(out):
                for f in files:
                    if f.endswith("counter_collection.csv"):
                        path = os.path.join(dirpath, f)
            if path is None:
                return None, "rocprofv3 --pmc %s wrote no counter file" % counter
            acc, cnt = {}, {}
            for row in csv.DictReader(open(path)):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                acc[name] = acc.get(name, 0.0) + float(row["Counter_Value"])
                cnt[name] = cnt.get(name, 0) + 1
            per[counter] = {n: acc[n] / cnt[n] for n in acc}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # k_fuse_tri<C, kind, exact, NV>: one instance per view count; the run-time-view kernels (k_fuse_tri_any / _wide / k_fuse_texel_multi) have one
    cands = [n for n in per["FETCH_SIZE"] if kernel_prefix in n and n in per["WRITE_SIZE"] and "_big" not in n]
    if not cands:
        return None, "no launch of %s in both counter files (FETCH_SIZE pass: %s; WRITE_SIZE pass: %s)" % (
            kernel_prefix, sorted(n for n in per["FETCH_SIZE"] if "fuse" in n)[:4], sorted(n for n in per["WRITE_SIZE"] if "fuse" in n)[:4])
    by_nv, names = {}, {}
    for n in cands:
        # (only k_fuse_tri<C, kind, exact, NV> carries its views per launch as its last template argument; k_fuse_tri_wide_list<kind, chunks> does not)
        tail = n.rstrip(">").rsplit(",", 1)[-1].strip() if (n.endswith(">") and "k_fuse_tri<" in n) else ""
        nv = int(tail) if tail.isdigit() and int(tail) in (1, 2, 4, 8) else 0
        t = int((2.0 * per["FETCH_SIZE"][n] + per["WRITE_SIZE"][n]) * 1024)
        if nv not in by_nv or t > by_nv[nv]:
            by_nv[nv], names[nv] = t, n
    top = max(by_nv, key=lambda k: by_nv[k])
    return by_nv, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --workload %s --steps %d "
                   "--warmup %d --no-group-pipeline`, kernel %s: (2 x %.0f + %.0f) KiB per launch" % (
                       workload, steps, warmup, names[top], per["FETCH_SIZE"][names[top]], per["WRITE_SIZE"][names[top]]))


def cpu_baseline(workload, budget_s=20.0):
    """The CPU oracle (a port of the reference's CPU fusion, include/semantic_meshes/fusion/Mesh.h:90-106,
    plus a CPU rasteriser -- the reference renders on CUDA only) timed on this box's host cores on a bounded
    sample of the same workload.  Test infrastructure used as the checker/baseline, never as the product."""
    from oracle import oracle
    cores = os.cpu_count() or 1
    oracle.set_threads(cores)
    oracle.set_accum_double(False)
    mesh, cams, C = synth.scene(workload)
    P = len(mesh.faces)
    r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg = oracle.OracleAggregator(P, C)
    W, H = cams[0].resolution
    probs = oracle.synth_probs(W * H, C, synth.probs_seed(1, 0)).reshape(W, H, C)
    done, t0 = 0, time.perf_counter()
    while done < len(cams):
        idx, _ = r.render(cams[done])
        agg.add(idx, probs)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    # the same sample with the one obvious CPU optimisation (dense parallel histogram instead of the serial std::map)
    # at the thread count that suits this host best: reported beside the faithful port so that the ratio is not only
    # against the reference's serial step (256 threads contending for one histogram are slower than 32)
    oracle.set_fast_histogram(True)
    agg2 = oracle.OracleAggregator(P, C)

    def run_views(nthreads, first, count):
        oracle.set_threads(nthreads)
        t = time.perf_counter()
        for k in range(first, first + count):
            idx, _ = r.render(cams[k % len(cams)])
            agg2.add(idx, probs)
        return count / (time.perf_counter() - t)

    candidates = sorted({c for c in (16, 32, 64, 128, cores) if c <= cores})
    run_views(candidates[0], 0, 1)   # page in
    rates = {c: run_views(c, 1, 2) for c in candidates}
    best = max(rates, key=rates.get)
    t1 = time.perf_counter()
    done2 = 0
    while time.perf_counter() - t1 < budget_s / 3:
        run_views(best, 3 + done2, 1)
        done2 += 1
    dt2 = time.perf_counter() - t1
    oracle.set_fast_histogram(False)
    # cfg1 (10 000 triangles, four 640 x 480 views, five classes -- the reference's own CPU-runnable case) in full
    cfg1 = None
    try:
        m1, c1, C1 = synth.scene("cfg1")
        r1 = oracle.OracleRenderer(m1.vertices, m1.faces)
        W1, H1 = c1[0].resolution
        p1 = [oracle.synth_probs(W1 * H1, C1, synth.probs_seed(1, k)).reshape(W1, H1, C1) for k in range(len(c1))]
        cfg1 = {}
        for label, nt, fast in (("port", cores, False), ("optimised_cpu", best, True)):
            oracle.set_threads(nt)
            oracle.set_fast_histogram(fast)
            a1 = oracle.OracleAggregator(len(m1.faces), C1)
            t = time.perf_counter()
            for k, cam in enumerate(c1):
                a1.add(r1.render(cam)[0], p1[k])
            a1.get()
            d1 = time.perf_counter() - t
            cfg1[label] = {"views_per_s": round(len(c1) / d1, 2), "seconds": round(d1, 4), "cores": nt}
        cfg1["what"] = "cfg1 in full: %d views at %dx%d, %d triangles, %d classes (render + add + one get)" % (len(c1), W1, H1, len(m1.faces), C1)
    except Exception as e:
        cfg1 = {"error": str(e)[:160]}
    oracle.set_fast_histogram(False)
    oracle.set_threads(1)
    return {"value": round(done / dt, 3), "unit": "views/s", "cores": cores, "kind": "port", "cfg1_full": cfg1,
            "sample": "%d of the %s views (render + add), %d OpenMP threads, %.1f s" % (done, workload, cores, dt),
            "optimised_cpu": {"value": round(done2 / dt2, 3), "unit": "views/s", "cores": best,
                              "what": "same port with a dense parallel histogram instead of the reference's serial std::map, "
                                      "best of %s threads" % candidates,
                              "sample": "%d views, %.1f s" % (done2, dt2)}}


def host_path(renderer, agg, cams, W, H, C, device, views=6):
    """The reference's usual calling convention (python/scripts/colorize_cityscapes_mesh.py:65-67): render(), then
    add(DEVICE indices, HOST numpy probs) -- every view's class vectors cross PCIe (python/semantic_meshes/include/Common.h:14-21
    accepts HOST or DEVICE).  Pageable numpy memory, and page-locked memory from the library.  Never `value`."""
    from semantic_meshes_amd.device import pinned_empty
    src = np.asarray(synth.device_probs(W, H, C, synth.probs_seed(7, 0), 0.0, device))
    out = {}
    for label in ("pageable", "pinned"):
        if label == "pinned":
            host = pinned_empty((W, H, C), np.float32)
            host[...] = src
        else:
            host = src
        idx, _ = renderer.render(cams[0])
        agg.add(idx, host)
        _lib.synchronize(device)
        t0 = time.perf_counter()
        for k in range(views):
            idx, _ = renderer.render(cams[k % len(cams)])
            agg.add(idx, host)
        _lib.synchronize(device)
        dt = time.perf_counter() - t0
        out[label] = {"views_per_s": round(views / dt, 1), "pcie_GBps": round(views * host.nbytes / dt / 1e9, 1)}
        del host
    out["what"] = "render() + add(device indices, host float32 (W,H,C) probs): %.1f MB per view over PCIe, %d views" % (4e-6 * W * H * C, views)
    return out


def foreign_images(renderer, P, cams, probs, W, H, C, T_mean, device, views=8):
    """add() on index images the library did not render (the reference's add() takes any (W,H) image, Mesh.h:65-107; its
    harness reloads renders from an .npz cache, eval_scannet.py:168-185): device-resident COPIES of renders -- no render
    matches by identity, content matching off -- with device-resident probs.  Per-primitive records are rebuilt from the image
    (image_records.hip: one atomic per (primitive, strip) group, round 3) and fused in triangle order at every class count; the atomic
    scatter-add remains behind SMESH_ADD_RECORDS=0 (fusion.hip, kAddRecordsMinC).  Never `value`."""
    from semantic_meshes_amd.device import to_device
    images = [to_device(np.asarray(renderer.render(cams[k % len(cams)])[0]), device) for k in range(views)]
    agg = fusion.MeshAggregator(primitives=P, classes=C, device=device)
    keep = fusion._MeshAggregator.match_renders
    fusion._MeshAggregator.match_renders = False
    try:
        for rep in range(2):
            _lib.synchronize(device)
            t0 = time.perf_counter()
            for k, img in enumerate(images):
                agg.add(img, probs[k % len(probs)])
            _lib.synchronize(device)
            dt = (time.perf_counter() - t0) / views
    finally:
        fusion._MeshAggregator.match_renders = keep
    bytes_per_view = 4 * W * H + 4 * W * H * C + 8 * C * T_mean
    scatter_path, scatter_kernel = _lib.last_add_path(), _lib.last_fuse_kernel()
    # ... and the same images handed over as ONE batch (add_many: the record passes of the eight images in one launch each, one
    # k_fuse_tri<.., 8> launch): same sums, bit for bit
    batched = None
    try:
        keep2 = fusion._MeshAggregator.match_renders
        fusion._MeshAggregator.match_renders = False
        bagg = fusion.MeshAggregator(primitives=P, classes=C, device=device)
        pb = [probs[k % len(probs)] for k in range(views)]
        best = None
        for rep in range(3):
            _lib.synchronize(device)
            t0 = time.perf_counter()
            bagg.add_many(images, pb)
            _lib.synchronize(device)
            d = (time.perf_counter() - t0) / views
            best = d if best is None else min(best, d)
        fusion._MeshAggregator.match_renders = keep2
        batched = {"ms_per_view": round(1e3 * best, 4), "frac": round(bytes_per_view / best / 1e9 / HBM_PEAK_GBS, 4),
                   "path": _lib.last_add_path(),
                   "what": "add_many(%d device copies of renders, device probs): one call, best of 3, host-timed" % views}
        del bagg
    except Exception as e:
        batched = {"error": str(e)[:200]}
    # ... and the same copies with content matching ON, of renders that were exported (np.asarray seals a plane): the harness-shaped
    # case (eval_scannet.py:211-238).  Round 2 recognised them by checksum and fused on the renderer's records; since round 3 the
    # Python layer skips the checksum (a host read-back per call) wherever add() rebuilds the records from the image -- `path` says which
    matched = None
    try:
        nm = min(views, 6)                                                # the renderer keeps the records of its last six renders
        agg.reset()
        best = None
        for rep in range(2):
            planes = [renderer.render(cams[k % len(cams)])[0] for k in range(nm)]
            copies = [to_device(np.asarray(p), device) for p in planes]
            _lib.synchronize(device)
            t0 = time.perf_counter()
            kernels = set()
            for k, img in enumerate(copies):
                agg.add(img, probs[k % len(probs)])
                kernels.add(_lib.last_fuse_kernel())
            _lib.synchronize(device)
            dtm = (time.perf_counter() - t0) / nm
            best = dtm if best is None else min(best, dtm)
            del planes, copies
        matched = {"ms_per_view": round(1e3 * best, 4), "kernels": sorted(kernels),
                   "frac": round(bytes_per_view / best / 1e9 / HBM_PEAK_GBS, 4),
                   "path": _lib.last_add_path(),
                   "what": "add(device copy of an EXPORTED render, device probs), content matching allowed "
                           "(taken only where the image records are off: SMESH_ADD_RECORDS_MIN_C / texel renderers), %d views" % nm}
    except Exception as e:
        matched = {"error": str(e)[:200]}
    return {"matched_copies": matched, "batched": batched, "ms_per_view": round(1e3 * dt, 4), "views_per_s": round(1.0 / dt, 1), "path": scatter_path,
            "kernel": scatter_kernel, "frac": round(bytes_per_view / dt / 1e9 / HBM_PEAK_GBS, 4),
            "what": "add(device copy of a render, device probs), one call per view, %d views, host-timed" % views}


def spawn_ranks(n):
    """`python bench.py --gpus N` with N > 1 and no launcher (no RANK in the environment): this process becomes the launcher -- N
    ranks of this very command line, one per GPU (RANK = LOCAL_RANK = device, WORLD_SIZE = N, rendezvous on 127.0.0.1), exactly the
    environment `python -m torch.distributed.run --nproc-per-node N` would have given them.  Rank 0 prints the one JSON line on this
    process's stdout.  Fails loudly when the node has fewer than N GPUs (SMESH_BENCH_BACKEND=gloo, the test mode in which ranks share
    GPUs, excepted).  Returns the exit status for the whole job."""
    import signal
    import socket
    import subprocess
    backend = os.environ.get("SMESH_BENCH_BACKEND", "nccl")
    have = _lib.device_count()
    if have < 1:
        raise SystemExit("bench: no HIP device visible -- the product has no CPU path")
    if have < n and backend != "gloo":
        raise SystemExit("bench: --gpus %d but this node shows %d GPU(s); one rank per GPU (RCCL refuses two ranks on a device)" % (n, have))
    # a base port with the rendezvous port and the native bootstrap's range (MASTER_PORT + 317 ...) free right now
    port = None
    for _ in range(64):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            cand = s.getsockname()[1]
        if cand + 317 + 8 >= 65536:
            continue
        ok = True
        for p in (cand, cand + 317):
            try:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
                    s.bind(("127.0.0.1", p))
            except OSError:
                ok = False
        if ok:
            port = cand
            break
    if port is None:
        raise SystemExit("bench: no free rendezvous port on 127.0.0.1")
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), GROUP_RANK="0",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SMESH_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        # ranks other than 0 print nothing on stdout by contract; whatever they do print must not reach the driver's JSON parser
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=False))
    status = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 128 - rc
                    print("bench: rank %d exited with status %d; stopping the other ranks" % (procs.index(p), rc), file=sys.stderr, flush=True)
                    for q in pending:          # (exactly the processes started above, by PID)
                        q.send_signal(signal.SIGTERM)
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            if q.poll() is None:
                q.send_signal(signal.SIGTERM)
        status = 130
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cfg2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", action="store_true", help="also for workloads other than cfg2 (slow: the oracle at that size)")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--pmc", action="store_true", default=None,
                    help="measure roofline.traffic in this run: two extra short passes of this script under rocprofv3 --pmc (about a minute). "
                         "Default: on for a one-GPU run of a triangle-renderer workload when rocprofv3 is on PATH")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="take roofline.traffic from the committed PMC summary instead")
    ap.add_argument("--views-per-call", type=int, default=int(os.environ.get("SMESH_BENCH_VIEWS_PER_CALL", "8")),
                    help="views handed to the library per call (fuse_views; 1 = one fuse_view call per view)")
    ap.add_argument("--exchange-parts", type=int, default=int(os.environ.get("SMESH_BENCH_EXCHANGE_PARTS", "4")),
                    help="N > 1: the rank's last --held-views views are fused by accumulator row range and the all-reduce of every finished "
                         "range runs on the exchange stream beside the fusion of the next (1 = one all-reduce after the last view)")
    ap.add_argument("--held-views", type=int, default=int(os.environ.get("SMESH_BENCH_HELD_VIEWS", "24")),
                    help="N > 1: how many of the rank's last views are fused by row range (at most 32)")
    ap.add_argument("--group-pipeline", dest="group_pipeline", action="store_true", default=None,
                    help="the library's default since round 5: the rasteriser of group g+1 on a second stream beside the fusion of group g "
                         "(same kernels, same results, +6 %% views/s).  The fusion kernel's own duration -- the roofline divisor -- is then "
                         "measured in a short SERIALISED leg of the same run (roofline.note)")
    ap.add_argument("--no-group-pipeline", dest="group_pipeline", action="store_false",
                    help="one kernel at a time on the GPU throughout (what rounds 1-4 timed); the roofline comes from the timed region itself")
    ap.add_argument("--repeats", type=int, default=int(os.environ["SMESH_BENCH_REPEATS"]) if os.environ.get("SMESH_BENCH_REPEATS") else None,
                    help="the K-step timed region (barrier, K steps, exchange, barrier) is run this many times; `value` is the MEDIAN region, "
                         "config.repeats / value_min / value_max / region_ms say how far the others were.  Default: 7, and 25 for K <= 40")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be at least 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: be the launcher (VERDICT r5 next 2 -- a plain `python bench.py --gpus 8` used to run ONE rank and print n_gpus 1)
        raise SystemExit(spawn_ranks(args.gpus))
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, os.environ.get("WORLD_SIZE", "1")))
    if args.group_pipeline is not None:
        _lib.check(_lib.lib().smesh_set_option(b"group_pipeline", 1 if args.group_pipeline else 0))
    gp = ctypes.c_int64(0)
    _lib.check(_lib.lib().smesh_get_option(b"group_pipeline", ctypes.byref(gp)))
    pipelined = bool(gp.value)
    if args.steps is None:
        args.steps = {"cfg5": 24}.get(args.workload, 200)
    if args.repeats is None:
        # A region of twenty cfg2 views is 1.3 ms of device time, and the first eight or so such regions of a process run 4 - 6 % slower than the
        # rest (1.37 -> 1.305 ms; profiles/r06_steps20_regions.txt): with seven regions the median sat on that ramp.  Short regions are repeated more
        # often -- every region is still K timed steps between two barriers, all of them are listed in config.region_ms, the value is their median.
        args.repeats = 7 if args.steps > 40 else 25
    args.repeats = max(1, args.repeats)
    if args.warmup is None:
        args.warmup = {"cfg5": 4}.get(args.workload, 10)

    if args.pmc is None:
        import shutil
        args.pmc = (args.gpus == 1 and "RANK" not in os.environ and shutil.which("rocprofv3") is not None
                    and not os.environ.get("SMESH_BENCH_NO_PMC"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SMESH_BENCH_BACKEND=gloo (tests): the N > 1 code path of this script on a box with fewer GPUs than ranks -- the ranks share
    # GPUs (local_rank modulo the device count) and exchange through torch.distributed's gloo backend (RCCL refuses two ranks on one
    # device).  The driver's runs never set it.
    backend = os.environ.get("SMESH_BENCH_BACKEND", "nccl")
    device = local_rank % max(_lib.device_count(), 1) if backend == "gloo" else local_rank
    launched = world > 1 or "RANK" in os.environ   # through torch.distributed.run (or any launcher that sets RANK)
    comm, dist, allreduce_impl = None, None, "none"
    if launched:
        if os.environ.get("SMESH_ALLREDUCE", "native") != "torch" and backend != "gloo":
            try:
                comm = smcomm.Communicator.from_env(device)
                allreduce_impl = "native smesh_allreduce (RCCL on the library stream)"
            except Exception as e:   # every rank fails the same way (no librccl / bootstrap port): fall back together
                print("bench: native communicator unavailable (%s); using torch.distributed" % e, file=sys.stderr, flush=True)
                comm = None
        if comm is None:
            import torch
            import torch.distributed as dist
            torch.cuda.set_device(device)
            if backend == "gloo":
                dist.init_process_group("gloo", rank=rank, world_size=world)
                allreduce_impl = "torch.distributed gloo (host copies; test mode, ranks sharing GPUs)"
            else:
                # device_id: the communicator is bound to this GPU and created now, not inside the timed region
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
                allreduce_impl = "torch.distributed nccl (RCCL), in place on the accumulator"
            if dist.get_world_size() != world or dist.get_rank() != rank:
                raise SystemExit("bench: the process group spans rank %d of %d, the launcher said rank %d of %d"
                                 % (dist.get_rank(), dist.get_world_size(), rank, world))

    cfg = synth.CONFIGS[args.workload]
    W, H, C = cfg["width"], cfg["height"], cfg["classes"]
    mesh = synth.grid_mesh(cfg["a"], cfg["b"])
    total_views = args.steps + args.warmup
    # rank r fuses views [r*total, (r+1)*total) of an (N * total)-view ring: weak scaling, cameras from a closed form
    view_ids = [rank * total_views + i for i in range(total_views)]
    ring = max(cfg["views"], world * total_views)
    cams = [synth.ring_camera(k, ring, W, H) for k in view_ids]

    texels = bool(cfg.get("texels"))
    if texels:
        # render.texels(mesh, cameras): the texel resolution of a triangle comes from its largest projection over the workspace's
        # cameras (TexturedTriangleRenderer.h:87-127) -- here the config's whole ring, the same on every rank
        ctor_cams = [synth.ring_camera(k, cfg["views"], W, H) for k in range(cfg["views"])]
        renderer = render.texels(mesh, ctor_cams, cfg.get("texels_per_pixel", 0.1), device=device)
    else:
        renderer = render.triangles(mesh, device=device)
    P = renderer.getPrimitivesNum()
    agg = fusion.MeshAggregator(primitives=P, classes=C, device=device)
    agg.defer = False     # `--views-per-call 1` means one library call per view (the Python layer would otherwise group fuse_view calls by eight)

    # ---- inputs resident in HBM before the timed region: one distinct probs image per view -------------
    # (cfg2: 210 x 157.6 MB = 33 GB.  Larger workloads cycle through as many images as fit in ~120 GB of HBM.)
    # (SMESH_BENCH_PROBS_GB: the counter passes of `--pmc` run as child processes BESIDE this one, whose images stay resident -- they cycle
    # through 40 GB of images instead of 120, or cfg5's two processes would not fit the 288 GB together)
    probs_budget = 1e9 * float(os.environ.get("SMESH_BENCH_PROBS_GB", "120"))
    nbuf = max(1, min(total_views, int(probs_budget // (4.0 * W * H * C))))
    bufs = [synth.device_probs(W, H, C, synth.probs_seed(1, k), 0.0, device) for k in view_ids[:nbuf]]
    probs = [bufs[i % nbuf] for i in range(total_views)]
    _lib.synchronize(device)

    # distinct primitives touched per view (T of the algorithmic-bytes formula) and visible pixels, on a sample of views
    sample = list(range(args.warmup, total_views, max(1, args.steps // 8)))[:8]
    T, NV = [], []
    for i in sample:
        idx, _ = renderer.render(cams[i])
        u = np.unique(np.asarray(idx))
        T.append(int((u < P).sum()))
        NV.append(int((np.asarray(idx) < P).sum()))
        del idx
    T_mean, NV_mean = float(np.mean(T)), float(np.mean(NV))

    def barrier():
        """every rank's queued work has finished and every rank has arrived (one host synchronisation)"""
        if comm is not None:
            comm.barrier()            # a 1-element all-reduce behind everything on the library stream + its wait
        else:
            _lib.synchronize(device)
            if dist is not None:
                import torch
                dist.barrier()
                torch.cuda.synchronize(device)

    exchange = os.environ.get("SMESH_EXCHANGE", "allreduce")      # "reduce_scatter": opt-in, each rank ends up owning P / N rows
    if exchange not in ("allreduce", "reduce_scatter"):
        raise SystemExit("SMESH_EXCHANGE must be allreduce or reduce_scatter")
    owned = [0, P]

    def allreduce():
        if exchange == "reduce_scatter" and (comm is not None or dist is not None):
            owned[:] = smdist.reduce_scatter_raw(agg, comm=comm)
        elif comm is not None:
            comm.allreduce(agg)       # asynchronous: enqueued behind the last fusion kernel
        elif dist is not None:
            smdist.allreduce_raw(agg)

    # The exchange under the fusion (N > 1, all-reduce): the rank's last `held` views are fused by accumulator row range, and the
    # all-reduce of every range that is final goes to the exchange stream while the next range is fused -- the same ONE sum of the
    # same 4 P C bytes, in `parts` pieces, of which only the last is exposed.  (torch.distributed plumbing: the ranges are exchanged
    # synchronously -- through host copies under gloo; the row arithmetic is the same, the overlap is not there.)
    parts = max(1, min(64, args.exchange_parts))
    held = max(0, min(32, args.held_views, args.steps))
    ranged = launched and exchange == "allreduce" and parts > 1 and held > 0 and not texels
    exchange_host_s = [0.0]

    def exchange_rows(lo, hi):
        t = time.perf_counter()
        smdist.allreduce_rows_raw(agg, lo, hi, comm=comm)     # native: queued on the exchange stream, returns at once
        exchange_host_s[0] += time.perf_counter() - t

    def fuse_and_exchange(first, last):
        """views [first, last) and the exchange; returns the row ranges that were exchanged one by one (None: one exchange at the end)"""
        if not ranged:
            fuse_range(first, last)
            mark(1)
            allreduce()
            return None
        cut = max(first, last - held)
        fuse_range(first, cut)
        ranges = agg.fuse_views_ranged(renderer, cams[cut:last], probs[cut:last], nparts=parts, on_rows=exchange_rows)
        mark(1)                                                   # behind the last part's fusion kernels on the main stream
        _lib.check(_lib.lib().smesh_exchange_join(agg._h))         # the main stream waits (on the device) for the exchange stream
        return ranges

    def mark(i):
        _lib.check(_lib.lib().smesh_stream_mark(device, i))    # an event record on the library stream, no host wait

    def elapsed(i, j):
        ms = ctypes.c_double()
        _lib.check(_lib.lib().smesh_stream_mark_elapsed(device, i, j, ctypes.byref(ms)))
        return ms.value

    B = max(1, args.views_per_call)
    call_sizes = [max(1, int(v)) for v in os.environ.get("SMESH_BENCH_CALL_SIZES", "").split(",") if v.strip()]

    def fuse_range(first, last):
        """views [first, last) in order: one fuse_view call per view, or fuse_views on batches of B (the library then
        shares launches between the views of a group)"""
        if B == 1:
            for i in range(first, last):
                agg.fuse_view(renderer, cams[i], probs[i])
        elif call_sizes:
            i, k = first, 0
            while i < last:      # (experiment: SMESH_BENCH_CALL_SIZES=4,8,4,4 -- calls of these sizes in turn; profiles/r06_steps20_regions.txt)
                j = min(i + call_sizes[k % len(call_sizes)], last)
                agg.fuse_views(renderer, cams[i:j], probs[i:j])
                i, k = j, k + 1
        else:
            for i in range(first, last, B):
                j = min(i + B, last)
                agg.fuse_views(renderer, cams[i:j], probs[i:j])

    # One untimed call with a full batch before anything else: the library allocates the per-view state of a group (fragment
    # queues, projected vertices, index planes: ~240 MB per view slot at cfg2's size) the first time a group of that many views
    # arrives, and that must not fall into the timed region when W is smaller than the batch.
    if B > 1:
        prime = [i % total_views for i in range(B)]
        agg.fuse_views(renderer, [cams[i] for i in prime], [probs[i] for i in prime])
        agg.fuse_view(renderer, cams[0], probs[0])     # a trailing odd view takes the one-view kernels: load them too
        _lib.synchronize(device)
    mark(0)
    if ranged:
        # untimed: the held views' record sets and index planes are allocated, RCCL builds its channels for the ranges' sizes
        fuse_and_exchange(0, min(total_views, max(args.warmup, held)))
    else:
        fuse_range(0, args.warmup)
        allreduce()    # untimed: RCCL builds its channels for this message size on first use
    barrier()
    agg.reset()
    exchange_host_s[0] = 0.0
    _lib.check(_lib.lib().smesh_profile_reset(device))
    prof_mask = 0xFF if os.environ.get("SMESH_BENCH_PROFILE_ALL") else ((1 << _lib.PROF_FUSE_SCATTER) | (1 << _lib.PROF_EXCHANGE))
    if os.environ.get("SMESH_BENCH_NO_PROFILE"):   # experiment: what do the HIP events around the kernel cost?
        prof_mask = 0
    if ranged and not os.environ.get("SMESH_BENCH_PROFILE_ALL"):
        # N > 1 with the exchange under the fusion: the held views' fusion is cut in `parts` x groups launches, and an event pair around
        # each of them keeps the stream idle ~10 us (measured: 12 pairs = 0.1 ms of a 1.7 ms region).  Only the collectives are timed here;
        # the kernel's roofline is the N = 1 line's.
        prof_mask = 1 << _lib.PROF_EXCHANGE
    # With the group pipeline the rasteriser of the next group runs BESIDE a fusion launch, so a launch's duration inside the timed
    # regions is not the kernel's own: the regions run without event pairs around the fusion launches (which also spares them the
    # ~6 us of stream idle time per event), and the kernel is timed in a serialised leg afterwards (below).
    serial_leg = pipelined and not launched and prof_mask != 0
    region_mask = (prof_mask & ~(1 << _lib.PROF_FUSE_SCATTER)) if serial_leg else prof_mask
    # HIP events on the library's stream around every 8th launch of the dominant kernel when every view is its own call (an event
    # pair costs ~4 us of stream time = 5 % of a cfg2 view); with fuse_views around the fusion launches of every THIRD group, the
    # first one included (measured: the two events of a group keep the stream idle for 2 x 6 us = 2 % of a group of eight cfg2
    # views; bracketing all of them lowered `value` from 14.3 k to 14.0 k views/s).  The library counts the launches and views
    # inside the bracketed regions itself (smesh_profile_read_ex), so the averages below are over exactly the regions that were timed.
    prof_every = int(os.environ.get("SMESH_BENCH_PROFILE_EVERY", "8" if B == 1 else "3"))
    if ranged or serial_leg:
        prof_every = 1      # (ranged: the fusion launches of a held group are cut in `parts` regions; serialised leg: every launch counts)
    elif B > 1 and args.steps <= 40 and "SMESH_BENCH_PROFILE_EVERY" not in os.environ:
        prof_every = 1      # short runs (the driver's --steps 20: three groups): every group, so that the average is over >= 3 launches
    _lib.check(_lib.lib().smesh_profile_sample_every(device, prof_every))

    # ---- the timed region, `repeats` times: barrier | K steps (+ the exchange) | barrier.  `value` is the median region. ----
    regions = []          # per repeat: [dt, compute_ms, exposed_ms, -exposed_ms] of THIS rank
    ranges = None
    for rep in range(args.repeats):
        agg.reset()       # (untimed: every region fuses its K views into an empty accumulator, the last one's result is what get() returns)
        _lib.check(_lib.lib().smesh_profile_enable(device, region_mask))
        barrier()
        _lib.synchronize(device)
        t0 = time.perf_counter()
        mark(0)
        ranges = fuse_and_exchange(args.warmup, total_views)     # (records mark 1 behind the last fusion kernel)
        mark(2)
        barrier()                      # the only host synchronisation of the timed region
        dt_rep = time.perf_counter() - t0
        _lib.check(_lib.lib().smesh_profile_enable(device, 0))
        # where this rank's device time went on the MAIN stream: marks 0 -> 1 = its views (render + fuse), 1 -> 2 = the part of the
        # exchange that nothing hid (its own transfers AND the wait for the slowest rank to arrive; with the exchange under the fusion:
        # what was left of the collectives when the last range had been fused).  Over all ranks: the max of each.
        c_ms, e_ms = elapsed(0, 1), elapsed(1, 2)
        regions.append([dt_rep, c_ms, e_ms, -e_ms])
    # the collectives themselves (HIP events around them on the stream they run on): equals the exposed part when nothing overlaps
    exchange_ms = prof_read(device, _lib.PROF_EXCHANGE)[0] / args.repeats if (comm is not None and region_mask) else None
    if ranged and comm is None:
        exchange_ms = 1e3 * exchange_host_s[0] / args.repeats                # (torch plumbing: host-timed, synchronous)
    flat = [v for r in regions for v in r] + [exchange_ms if exchange_ms is not None else 0.0]
    if comm is not None:
        flat_max = []
        for i in range(0, len(flat), 60):                      # (smesh_comm_allreduce_f64 takes at most 64 values)
            flat_max += comm.reduce_scalars(flat[i:i + 60], "max")
    elif dist is not None:
        import torch
        t = torch.tensor(flat, dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda:%d" % device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        flat_max = [float(v) for v in t.tolist()]
    else:
        flat_max = list(flat)
    regions_max = [flat_max[4 * i:4 * i + 4] for i in range(args.repeats)]
    order = sorted(range(args.repeats), key=lambda i: regions_max[i][0])
    med = order[(args.repeats - 1) // 2]                            # the median region (the lower one of an even count)
    dt, compute_ms_max, exposed_ms_max = regions_max[med][0], regions_max[med][1], regions_max[med][2]
    exposed_ms_min = -regions_max[med][3]
    dt_min, dt_max = regions_max[order[0]][0], regions_max[order[-1]][0]
    compute_ms, exposed_ms = regions[med][1], regions[med][2]
    if exchange_ms is None:
        exchange_ms, exchange_ms_max = exposed_ms, exposed_ms_max
    else:
        exchange_ms_max = flat_max[-1]

    # ---- the serialised leg: the fusion kernel's own duration (roofline), one kernel at a time on the GPU ----
    leg_views = 0
    if serial_leg:
        _lib.check(_lib.lib().smesh_set_option(b"group_pipeline", 0))
        _lib.check(_lib.lib().smesh_profile_reset(device))
        _lib.check(_lib.lib().smesh_profile_sample_every(device, 1))
        leg_views = min(args.steps, 64)
        leg_first = args.warmup
        fuse_range(leg_first, leg_first + min(B, leg_views))         # (untimed: the first serial group behind pipelined ones)
        _lib.synchronize(device)
        _lib.check(_lib.lib().smesh_profile_enable(device, prof_mask & ~(1 << _lib.PROF_EXCHANGE)))
        fuse_range(leg_first, leg_first + leg_views)
        _lib.synchronize(device)
        _lib.check(_lib.lib().smesh_profile_enable(device, 0))
        _lib.check(_lib.lib().smesh_set_option(b"group_pipeline", 1))
        # (the accumulator now holds the leg's views too: the result checks below only count annotated primitives)

    t1 = time.perf_counter()
    fused = agg.get() if exchange == "allreduce" else agg.get_rows(*owned)
    get_first_ms = 1e3 * (time.perf_counter() - t1)     # (the first get() of the process also allocates its page-locked landing buffer)
    annotated = int((fused.sum(axis=1) > 0.9).sum())
    if os.environ.get("SMESH_BENCH_DUMP") and rank == 0 and exchange == "allreduce":
        # (tests: a sample of the fused rows, to be compared with a single-process fusion of the same views)
        sample_rows = np.unique(np.linspace(0, max(P - 1, 0), 4096).astype(np.int64))
        np.savez(os.environ["SMESH_BENCH_DUMP"], rows=sample_rows, fused=fused[sample_rows])
    del fused
    t1 = time.perf_counter()
    fused = agg.get() if exchange == "allreduce" else agg.get_rows(*owned)     # ... which the next result of that size recycles
    get_ms = 1e3 * (time.perf_counter() - t1)
    del fused
    nranks_reported = world
    if comm is not None:
        rk, nr = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().smesh_comm_rank(comm._h, ctypes.byref(rk), ctypes.byref(nr)))
        nranks_reported = int(nr.value)        # what the RCCL communicator itself spans
    elif dist is not None:
        nranks_reported = dist.get_world_size()
    if nranks_reported != args.gpus:
        raise SystemExit("bench: --gpus %d but the process group spans %d rank(s)" % (args.gpus, nranks_reported))

    fuse_kernel = _lib.last_fuse_kernel()
    k_ms, k_regions, k_launches, k_views = prof_read(device, _lib.PROF_FUSE_SCATTER)
    entered = ctypes.c_uint64(0)       # regions of the dominant kernel the timed loop went through, bracketed or not
    _lib.check(_lib.lib().smesh_profile_regions(device, _lib.PROF_FUSE_SCATTER, ctypes.byref(entered)))
    k_entered = int(entered.value)
    hist_ms, hist_regions, _, _ = prof_read(device, _lib.PROF_FUSE_HIST)
    raster_ms, raster_regions, _, _ = prof_read(device, _lib.PROF_RASTER)

    if rank == 0:
        N = W * H
        F = len(mesh.faces)
        # idx + probs + accumulator read-modify-write of the T primitives touched
        bytes_per_view = 4 * N + 4 * N * C + 8 * C * T_mean
        # what a triangle-order kernel has to move at least: the index plane, the class vectors of the VISIBLE pixels only,
        # the per-triangle records, the touched accumulator rows both ways
        needed_per_view = 4 * N + 4 * C * NV_mean + 16 * F + 8 * C * T_mean
        t_launch = k_ms * 1e-3 / max(k_launches, 1)                  # average duration of one launch of the dominant kernel
        views_per_launch = k_views / max(k_launches, 1)
        # which launches those were: a call of n views is cut into launches of 8 / 4 / 2 / 1 views, largest first (the library's
        # rule, smesh_fuse_views); cross-checked against the library's own counters
        mix = {}
        if B > 1 and prof_mask and not ranged:
            cap = 8     # smesh_aggregator_max_fused_views (Mul with 41 .. 48 classes: 2 -- not a bench workload)
            cap = min(cap, int(os.environ.get("SMESH_FUSE_VIEWS", "8")))
            # the calls whose fusion launches were bracketed: those of the serialised leg, or of every timed region
            first_call, end_call, times = ((args.warmup, args.warmup + leg_views, 1) if serial_leg else (args.warmup, total_views, args.repeats))
            for call, i in enumerate(range(first_call, end_call, B)):
                if call % prof_every:
                    continue           # (not one of the bracketed calls)
                left = min(i + B, end_call) - i
                while left:
                    nv = 1
                    while nv * 2 <= min(cap, left):
                        nv *= 2
                    mix[nv] = mix.get(nv, 0) + times
                    left -= nv
            if sum(mix.values()) != k_launches or sum(k * v for k, v in mix.items()) != k_views:
                mix = {}
        timed = k_launches > 0 and k_ms > 0     # (N > 1 with the exchange under the fusion: the fusion launches are not bracketed)
        achieved = bytes_per_view * k_views / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        achieved_needed = needed_per_view * k_views / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # roofline.traffic: HBM bytes per launch from the PMC counters, for the instance that fuses the most views per launch;
        # frac_traffic: the counter bytes of ALL the timed launches (each instance's own bytes x how often it was timed; an
        # instance the counter passes did not see: the largest one's bytes scaled by views) over the time those launches took --
        # round 4 divided the eight-view instance's bytes by the mean duration of an {8, 8, 4} mix
        traffic, traffic_source, traffic_by_nv, traffic_timed = None, None, None, None
        vpl_int = int(round(views_per_launch)) if views_per_launch else 1
        if args.pmc and world == 1:
            try:
                sub_steps, sub_warm = (args.steps, args.warmup) if args.steps <= 40 else (16, 8)
                traffic_by_nv, traffic_source = pmc_traffic(args.workload, fuse_kernel, sub_steps, sub_warm)
            except Exception as e:
                traffic_by_nv, traffic_source = None, "pmc passes failed: %s" % str(e)[:160]
        if traffic_by_nv:
            top = max(traffic_by_nv)                      # views per launch of the largest instance (0: a run-time-view kernel)
            traffic = traffic_by_nv[top]
            if top == 0 or not mix:
                per_view = traffic / max(views_per_launch if top == 0 else top, 1)
                traffic_timed = per_view * k_views
            else:
                traffic_timed = sum(cnt * (traffic_by_nv[nv] if nv in traffic_by_nv else traffic * nv / top) for nv, cnt in mix.items())
        if traffic is None:
            # not measured in this run: the committed PMC summary of an earlier round, for the kernel instance that ran
            note = traffic_source
            tpath = os.path.join(ROOT, "profiles", "fusion_traffic.json")
            if os.path.exists(tpath):
                try:
                    tkey = "%s:%s" % (args.workload, fuse_kernel) + ("_x8" if views_per_launch > 6 else "_pair" if views_per_launch > 1.5 else "")
                    entry = json.load(open(tpath)).get(tkey)
                    if entry:
                        traffic = entry.get("hbm_bytes_per_launch")
                        vpl_entry = 8 if tkey.endswith("_x8") else 2 if tkey.endswith("_pair") else 1
                        traffic_timed = traffic / vpl_entry * k_views
                        traffic_source = ("profiles/fusion_traffic.json [%s] (committed PMC passes of an EARLIER round; `--pmc` measures it in the run%s)"
                                          % (tkey, "; this run's passes: " + note if note else ""))
                except Exception:
                    traffic = None
            if traffic is None and note:
                traffic_source = note
        out = {
            "metric": ("views/sec fused (1080p, 19 classes, 1M-tri mesh)" if args.workload == "cfg2" else
                       "views/sec fused (%s: %dx%d, %d classes, %d triangles%s)" % (args.workload, W, H, C, F,
                                                                                   ", %d texel primitives" % P if texels else "")),
            "value": round(world * args.steps / dt, 2),            # the MEDIAN of `repeats` timed regions
            "unit": "views/s",
            "n_gpus": nranks_reported,          # what the communicator / process group spans (== --gpus, checked above)
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s: %d-triangle grid mesh%s, %d views/GPU at %dx%d, %d classes, probs resident in HBM"
                                   % (args.workload, F, " as %d texel primitives" % P if texels else "", args.steps, W, H, C),
                       "views_per_call": B,
                       "group_pipeline": pipelined,
                       "repeats": args.repeats,
                       "value_min": round(world * args.steps / dt_max, 2), "value_max": round(world * args.steps / dt_min, 2),
                       "value_spread": round((dt_max - dt_min) / dt, 4),
                       "region_ms": [round(1e3 * r[0], 3) for r in regions_max],
                       "sharding": "views dp%d, one RCCL %s of float32[P,C]" % (world, exchange.replace("_", "-")),
                       "allreduce": allreduce_impl, "exchange": exchange, "nranks": nranks_reported,
                       # the exchange moves the raw accumulator once: ring all-reduce 2 (N-1)/N x, reduce-scatter (N-1)/N x per link
                       "allreduce_bytes": int(4 * P * C) if launched else 0,
                       # compute_ms + exchange_exposed_ms ~ timed_region_ms; exchange_ms = the collectives' own duration (under the fusion
                       # when exchange_parts > 1: then exchange_ms > exchange_exposed_ms is the point)
                       "compute_ms": round(compute_ms_max, 3), "exchange_ms": round(exchange_ms_max, 3),
                       "exchange_exposed_ms": round(exposed_ms_max, 3),
                       "exchange_exposed_ms_fastest_rank": round(exposed_ms_min, 3),
                       "exchange_parts": (len(ranges) if ranges else 1), "held_views": (held if ranges else 0),
                       "exchange_row_ranges": ranges,
                       "rank0": {"compute_ms": round(compute_ms, 3), "exchange_ms": round(exchange_ms, 3),
                                 "exchange_exposed_ms": round(exposed_ms, 3)},
                       "timed_region_ms": round(1e3 * dt, 3),
                       "host_syncs_in_timed_region": 1 if (comm is not None or dist is None) else 3 + (parts if ranged else 1),
                       "get_ms": round(get_ms, 2), "get_first_ms": round(get_first_ms, 2), "annotated_primitives": annotated},
            # frac_needed first: the fraction by the bytes the kernel HAS to move (visible pixels' class vectors, records, touched
            # rows once per launch); `frac` is SURVEY.md 8(d)'s formula, which also charges the class vectors of background
            # pixels and a row round trip per view that the kernel does not perform -- it flatters
            "roofline": {"frac_needed": round(achieved_needed / HBM_PEAK_GBS, 4) if timed else None,
                         "kernel": KERNEL_NOTES.get(fuse_kernel, fuse_kernel),
                         "note": (("N > 1 with the exchange under the fusion: the fusion launches are not bracketed with events here (an event pair "
                                   "around each of exchange_parts x groups launches costs what the overlap saves); the kernel's roofline is the "
                                   "N = 1 line's") if ranged else
                                  ("kernel durations from a SERIALISED leg of this run (group pipeline off, %d views right after the timed regions, every "
                                   "fusion launch bracketed with HIP events on the library stream): in the timed regions the rasteriser of the next group "
                                   "runs beside each fusion launch, whose duration is then not the kernel's own.  us_per_view <= ms_per_step still holds."
                                   % leg_views) if serial_leg else None),
                         "bound": "hbm",
                         "achieved": round(achieved, 1) if timed else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if timed else None, "traffic": traffic, "traffic_source": traffic_source,
                         "frac_traffic": (round(traffic_timed / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if (traffic_timed and timed) else None),
                         "traffic_by_views_per_launch": ({str(k): v for k, v in sorted(traffic_by_nv.items(), reverse=True)} if traffic_by_nv else None),
                         "algorithmic_bytes_per_view": int(bytes_per_view),
                         "algorithmic_bytes_per_launch": int(bytes_per_view * views_per_launch),
                         "needed_bytes_per_view": int(needed_per_view),
                         "views_per_launch": (int(views_per_launch) if views_per_launch == int(views_per_launch)
                                              else round(views_per_launch, 3)),
                         "launches_by_views": {str(k): v for k, v in sorted(mix.items(), reverse=True)} or None,
                         "avg_launch_us": round(1e6 * t_launch, 2) if timed else None,
                         "us_per_view": round(1e3 * k_ms / max(k_views, 1), 3) if timed else None,
                         "launches_timed": k_launches, "views_timed": k_views, "regions_timed": k_regions,
                         "regions_in_timed_loop": k_entered,
                         "distinct_primitives_per_view": int(T_mean), "visible_pixels_per_view": int(NV_mean),
                         "other_kernels_us_per_view": ({"histogram+pixel_weights": round(1e3 * hist_ms / max(args.steps, 1), 2),
                                                        "raster": round(1e3 * raster_ms / max(raster_regions, 1) / max(1, min(B, 8)), 2)}
                                                       if hist_regions or raster_regions else None)},
        }
        if world == 1 and not args.no_host_path:
            # side legs: reported beside the headline, never allowed to take it down
            try:
                out["host_path"] = host_path(renderer, agg, cams, W, H, C, device, views=6 if args.workload != "cfg5" else 2)
            except Exception as e:
                out["host_path"] = {"error": str(e)[:200]}
            if not texels:
                try:
                    out["foreign_images"] = foreign_images(renderer, P, cams, probs, W, H, C, T_mean, device,
                                                           views=8 if args.workload != "cfg5" else 4)
                except Exception as e:
                    out["foreign_images"] = {"error": str(e)[:200]}
        if world == 1 and ((args.workload == "cfg2" and not args.no_cpu_baseline) or args.cpu_baseline):
            try:
                out["cpu_baseline"] = cpu_baseline(args.workload)
            except Exception as e:      # (the headline line is printed whatever happens to the CPU leg)
                out["cpu_baseline"] = {"value": None, "unit": "views/s", "cores": 0, "kind": "port", "sample": "failed: " + str(e)[:200]}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
