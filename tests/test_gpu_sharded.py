"""The N > 1 path with the PRODUCT on it: two processes share the one GPU of the test box, each
fuses its shard of the views with HIP renderers / aggregators (`fuse_views_sharded`), the raw accumulators are exchanged
through torch.distributed's gloo backend (RCCL refuses two ranks on one device: "Duplicate GPU detected"), and every rank's
`get()` equals the single-process fusion of all views.  The oracle is the checker of the single-process job only."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SMESH_ROOT"])
sys.path.insert(0, os.path.join(os.environ["SMESH_ROOT"], "tests"))
import semantic_meshes_amd as sm
from semantic_meshes_amd import _lib, distributed as smdist, synth
import torch.distributed as dist
from helpers import small_scene, assert_fused_close

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
assert _lib.lib().smesh_backend() == b"hip-gfx950"
mesh, cams = small_scene(60, 30, 320, 240, views=8)
P, C = len(mesh.faces), 19
renderer = sm.render.triangles(mesh)
repeats = int(os.environ.get("SMESH_SHARDED_REPEATS", "1"))
dump_dir = os.environ["SMESH_DUMP_DIR"]

def probs_of_view(k):
    W, H = cams[k].resolution
    return synth.device_probs(W, H, C, synth.probs_seed(3, k), 0.05, 0)

# Self-diagnosis (VERDICT r4 next #1): every host-side exchange of a row range is bracketed by snapshots of the rows it moves --
# this rank's partial sums going in, the sums over ranks coming out -- which every rank writes per leg; a leg that fails also writes
# them, with what it compared, to SMESH_DUMP_DIR/FAILED_rank<r>_<kind>_<leg>_<repeat>.npz before it raises.  The parent then adds
# the ranks' partial sums up itself (diagnose_sharded_dumps): partial sums that do not add up to the single-process job -> a rank's
# FUSION lost or doubled something; they do, and the rows coming out of the exchange differ -> the EXCHANGE; both fine -> get() /
# get_rows().
snap = []
_exchange = smdist.allreduce_rows_raw
def _spied(aggregator, row_lo, row_hi, group=None, comm=None):
    pre = aggregator.get_raw_rows(row_lo, row_hi)
    out = _exchange(aggregator, row_lo, row_hi, group, comm)
    snap.append((int(row_lo), int(row_hi), pre, aggregator.get_raw_rows(row_lo, row_hi)))
    return out
smdist.allreduce_rows_raw = _spied

def leg(name, kind, rep, check, **arrays):
    out = dict(arrays, repeat=np.asarray(rep))
    for i, (lo, hi, pre, post) in enumerate(snap):
        out["x%d_range" % i] = np.asarray([lo, hi]); out["x%d_pre" % i] = pre; out["x%d_post" % i] = post
    del snap[:]
    # every rank leaves what it exchanged in this leg (overwritten by the next repeat): when ONE rank fails, its peers stop at the next
    # collective with this leg's file still in place, and the parent adds the partial sums of all ranks up
    np.savez(os.path.join(os.environ["SMESH_SNAP_DIR"], "rank%d_%s_%s.npz" % (rank, kind, name)), **out)
    try:
        check()
    except AssertionError as e:
        path = os.path.join(dump_dir, "FAILED_rank%d_%s_%s_%d.npz" % (rank, kind, name, rep))
        np.savez(path, **out)
        print("SHARDED-MISMATCH rank %d kind %s leg %s repeat %d -> %s: %s" % (rank, kind, name, rep, path, e), flush=True)
        raise

out = {}
for rep in range(repeats):
  for kind in ("sum", "summax", "mul"):
    whole = sm.fusion.MeshAggregator(P, C, kind)
    whole.fuse_views(renderer, cams, [probs_of_view(k) for k in range(len(cams))])
    want, want_raw = whole.get(), whole.get_raw()
    # the default: the rank's views fused by accumulator row range, every finished range exchanged at once (three ranges here)
    agg = sm.fusion.MeshAggregator(P, C, kind)
    _, rows = smdist.fuse_views_sharded(renderer, agg, cams, probs_of_view, contiguous=(kind != "summax"), nparts=3)
    assert rows == (0, P)
    kernel = _lib.last_fuse_kernel()
    assert kernel.startswith("k_fuse_tri"), kernel            # the HIP triangle-order path ran on this rank's shard
    got, got_raw = agg.get(), agg.get_raw()
    # partial float32 sums added once instead of eight terms in order: 1e-5 for every aggregator (Mul's (hi, lo) pairs travel as float64)
    def ranged():
        assert_fused_close(got, want, rtol=1e-5)
        np.testing.assert_allclose(got_raw, want_raw, rtol=1e-5, atol=1e-5)
    leg("ranged", kind, rep, ranged, got=got, want=want, got_raw=got_raw, want_raw=want_raw)
    assert (want.sum(axis=1) > 0.5).sum() > P // 3
    # ... and ONE all-reduce after the last view (nparts = 1): the same sums
    agg1 = sm.fusion.MeshAggregator(P, C, kind)
    smdist.fuse_views_sharded(renderer, agg1, cams, probs_of_view, contiguous=(kind != "summax"), nparts=1)
    got1, got1_raw = agg1.get(), agg1.get_raw()
    def single():
        assert_fused_close(got1, want, rtol=1e-5)
        # (per row the same float32 additions in the same order; the sum over ranks may associate differently per range)
        np.testing.assert_allclose(got1_raw, got_raw, rtol=2e-6, atol=1e-6)
    leg("one_exchange", kind, rep, single, got=got1, want=want, got_raw=got1_raw, want_raw=want_raw)
    # opt-in exchange: this rank normalises its own slice of rows
    agg2 = sm.fusion.MeshAggregator(P, C, kind)
    _, (lo, hi) = smdist.fuse_views_sharded(renderer, agg2, cams, probs_of_view, exchange="reduce_scatter")
    assert (lo, hi) == smdist.owned_rows(P, rank, world)
    got2 = agg2.get_rows(lo, hi)
    def scattered():
        assert_fused_close(got2, want[lo:hi], rtol=1e-5)
    leg("reduce_scatter", kind, rep, scattered, got=got2, want=want[lo:hi], rows=np.asarray([lo, hi]), want_raw=want_raw,
        got_raw=agg2.get_raw_rows(0, P))
    out[kind] = got
np.savez(os.environ["SMESH_OUT"], **out)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def run_sharded_workers(tmp_path, world, repeats=1, timeout=600):
    """Launches WORKER as `world` ranks sharing the box's GPU; returns (return codes, outputs, dump directory).  A rank whose leg
    fails leaves FAILED_rank<r>_<kind>_<leg>_<repeat>.npz in gpurun_out/sharded_dump (which comes back from the GPU box); the
    per-leg snapshots of ALL ranks are then copied beside it."""
    import glob
    import shutil
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(WORKER)
    dump_dir = os.path.join(ROOT, "gpurun_out", "sharded_dump")
    os.makedirs(dump_dir, exist_ok=True)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMESH_ROOT=ROOT, SMESH_OUT=os.path.join(tmp_path, "rank%d.npz" % rank), OMP_NUM_THREADS="1",
                   SMESH_DUMP_DIR=dump_dir, SMESH_SNAP_DIR=str(tmp_path), SMESH_SHARDED_REPEATS=str(repeats))
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    deadline = None
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout if deadline is None else 20)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        if p.returncode != 0:
            deadline = True     # (a failed rank leaves its peers blocked in the next collective: no point in waiting `timeout` for each)
        outs.append(out.decode(errors="replace"))
    codes = [p.returncode for p in procs]
    if any(codes):
        for path in glob.glob(os.path.join(tmp_path, "rank*_*_*.npz")):
            shutil.copy(path, dump_dir)
    return codes, outs, dump_dir


def diagnose_sharded_dumps(dump_dir):
    """What the dumps of a failed run say, one line per failed leg: which stage produced the wrong rows -- a rank's fusion (the
    partial sums of the ranks do not add up to the single-process sums), the exchange (they do, what came out of it differs) or the
    read-out (both fine)."""
    import glob
    lines = []
    for path in sorted(glob.glob(os.path.join(dump_dir, "FAILED_rank*.npz"))):
        name = os.path.basename(path)[len("FAILED_"):-len(".npz")]
        rank_s, kind, rest = name.split("_", 2)
        legname, rep = rest.rsplit("_", 1)
        d = np.load(path)
        peers = [np.load(q) for q in sorted(glob.glob(os.path.join(dump_dir, "rank*_%s_%s.npz" % (kind, legname))))]
        peers = [q for q in peers if int(q["repeat"]) == int(rep)]
        verdict = ["%d peer snapshots of repeat %s" % (len(peers), rep)]
        n = 0
        while "x%d_range" % n in d:
            lo, hi = (int(v) for v in d["x%d_range" % n])
            post, want = d["x%d_post" % n].astype(np.float64), d["want_raw"][lo:hi].astype(np.float64)
            tol = 1e-5 * np.abs(want) + 1e-5
            verdict.append("rows [%d, %d): %d rows out of the exchange differ from the single-process sums" % (lo, hi, (np.abs(post - want) > tol).any(axis=1).sum()))
            if peers and all("x%d_pre" % n in q for q in peers) and kind != "mul":
                total = sum(q["x%d_pre" % n].astype(np.float64) for q in peers)
                verdict.append("sum of the %d ranks' partial sums: %d rows differ from the single-process sums (FUSION if > 0), %d rows "
                               "differ from what the exchange returned (EXCHANGE if > 0)" % (
                                   len(peers), (np.abs(total - want) > tol).any(axis=1).sum(), (np.abs(total - post) > tol).any(axis=1).sum()))
            n += 1
        lines.append("%s: %s" % (os.path.basename(path), "; ".join(verdict)))
    return lines


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_on_one_gpu_hip_aggregators_equal_the_single_process_job(tmp_path, sm, oracle, world):
    """Two ranks, and cfg3's eight (one view per rank of the eight-view scene), sharing the test box's GPU."""
    codes, outs, dump_dir = run_sharded_workers(tmp_path, world)
    for rank, (code, out) in enumerate(zip(codes, outs)):
        assert code == 0, "rank %d failed:\n%s\n%s" % (rank, out[-3000:], "\n".join(diagnose_sharded_dumps(dump_dir)))
        assert "rank %d ok" % rank in out
    # every rank holds the same result (an all-reduce), and it is the oracle's fusion of all eight views
    from helpers import small_scene, assert_fused_close
    from semantic_meshes_amd import synth
    r0, r1 = np.load(os.path.join(tmp_path, "rank0.npz")), np.load(os.path.join(tmp_path, "rank%d.npz" % (world - 1)))
    mesh, cams = small_scene(60, 30, 320, 240, views=8)
    P, C = len(mesh.faces), 19
    oracle.set_accum_double(True)
    try:
        o_r = oracle.OracleRenderer(mesh.vertices, mesh.faces)
        for kind in ("sum", "summax", "mul"):
            assert np.array_equal(r0[kind], r1[kind])
            o_a = oracle.OracleAggregator(P, C, kind)
            for k, cam in enumerate(cams):
                W, H = cam.resolution
                o_a.add(o_r.render(cam)[0], oracle.synth_probs(W * H, C, synth.probs_seed(3, k), 0.05).reshape(W, H, C))
            assert_fused_close(r0[kind], o_a.get(), rtol=1e-5)
    finally:
        oracle.set_accum_double(False)


def test_native_reduce_scatter_world_one_owns_every_row(sm):
    """`smesh_reduce_scatter` with a one-rank RCCL communicator: the whole accumulator is this rank's slice and `get_rows`
    of it is `get()`; `get_rows` on an inner range equals the slice of `get()`."""
    from semantic_meshes_amd import comm as smcomm, synth
    from helpers import small_scene
    mesh, cams = small_scene(30, 15, 160, 120, views=2)
    P, C = len(mesh.faces), 7
    renderer = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    agg.fuse_views(renderer, cams, [synth.device_probs(160, 120, C, synth.probs_seed(5, k), 0.0, 0) for k in range(2)])
    want = agg.get()
    assert np.array_equal(agg.get_rows(8, 101), want[8:101])
    with pytest.raises(ValueError):
        agg.get_rows(3, 10)
    c = smcomm.Communicator(0, 0, 1, smcomm.Communicator.unique_id())
    lo, hi = c.reduce_scatter(agg)
    assert (lo, hi) == (0, P)
    assert np.array_equal(agg.get_rows(lo, hi), want)


@pytest.mark.parametrize("exchange", ["allreduce", "reduce_scatter"])
def test_bench_line_of_a_two_rank_run(tmp_path, sm, exchange):
    """bench.py exactly as the driver launches it for N = 2 (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`),
    on the one GPU of the test box: SMESH_BENCH_BACKEND=gloo lets the two ranks share it.  The N > 1 branch of the script -- view
    sharding, the exchange inside the timed region, barrier, max over ranks, the diagnosable fields -- runs and prints ONE JSON line."""
    import json
    port = _free_port()
    env = dict(os.environ, SMESH_BENCH_BACKEND="gloo", SMESH_EXCHANGE=exchange, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--workload", "cfg1", "--repeats", "7"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 12 / (d["ms_per_step"] * 12 * 1e-3)) < 0.02 * d["value"]          # whole-job views / max-over-ranks time
    assert cfg["nranks"] == 2 and cfg["exchange"] == exchange and cfg["allreduce_bytes"] == 4 * 10000 * 5
    assert cfg["compute_ms"] > 0 and cfg["exchange_ms"] > 0 and cfg["exchange_exposed_ms"] >= cfg["exchange_exposed_ms_fastest_rank"]
    assert cfg["compute_ms"] + cfg["exchange_exposed_ms_fastest_rank"] <= cfg["timed_region_ms"] * 1.05
    if exchange == "allreduce":     # the default: the last views by row range, every finished range exchanged at once
        assert cfg["exchange_parts"] == 4 and cfg["held_views"] == 12
        rr = cfg["exchange_row_ranges"]
        assert rr[0][0] == 0 and rr[-1][1] == 10000 and all(a[1] == b[0] for a, b in zip(rr, rr[1:]))
    else:
        assert cfg["exchange_parts"] == 1
    assert "cpu_baseline" not in d and ((d["roofline"]["frac"] or 0) > 0 or exchange == "allreduce")   # (cut fusion launches are not bracketed)


def test_bench_gpus_two_without_a_launcher_spawns_two_ranks(tmp_path, sm):
    """`python bench.py --gpus 2` started PLAINLY (no torch.distributed.run, no RANK in the environment -- the way the driver starts
    `--gpus 1`): the script is its own launcher (bench.spawn_ranks) and the line says two ranks -- n_gpus and config.nranks come from the
    process group, not from the flag.  (this command used to run one rank and print n_gpus 1.)  On the one-GPU box
    SMESH_BENCH_BACKEND=gloo lets the two ranks share the device; without it the script must refuse, loudly, to start two RCCL ranks
    on one GPU."""
    import json
    from semantic_meshes_amd import _lib
    env = dict(os.environ, SMESH_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SMESH_EXCHANGE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--workload", "cfg1", "--repeats", "7"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["nranks"] == 2 and d["steps"] == 12 and d["value"] > 0
    assert d["config"]["allreduce_bytes"] == 4 * 10000 * 5 and d["config"]["exchange_parts"] == 4
    if _lib.device_count() < 2:
        env.pop("SMESH_BENCH_BACKEND")
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
        assert out.returncode != 0 and "one rank per GPU" in out.stderr and '{"metric"' not in out.stdout, out.stdout[-500:] + out.stderr[-1500:]
    # a launcher whose world size contradicts --gpus is refused too
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", SMESH_BENCH_BACKEND="gloo")
    out = subprocess.run(cmd, env=env2, capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr, out.stderr[-1500:]


def test_fuse_views_ranged_equals_fuse_views_bit_for_bit(sm):
    """`fuse_views_ranged` (smesh_fuse_views_begin / _continue) cuts the job by accumulator row range: per row the same float32
    additions in the same order as `fuse_views` -- raw accumulators bit-equal for Sum, Summax and Mul, for part counts that do and do
    not divide the block count, eleven views (groups of 8 + 2 + 1), and a class count that takes k_fuse_tri_any (C = 53)."""
    from semantic_meshes_amd import synth
    from helpers import small_scene
    mesh, cams = small_scene(170, 81, 320, 240, views=11)   # (triangles of two or three pixels: no box over 8 x 8, whose float atomics have no order)
    P = len(mesh.faces)
    renderer = sm.render.triangles(mesh)
    for C in (19, 53):
        probs = [synth.device_probs(320, 240, C, synth.probs_seed(9, k), 0.05, 0) for k in range(len(cams))]
        for kind in ("sum", "summax", "mul"):
            whole = sm.fusion.MeshAggregator(P, C, kind)
            whole.fuse_views(renderer, cams, probs)
            want = whole.get_raw()
            for nparts in (1, 3, 4, 64):
                agg = sm.fusion.MeshAggregator(P, C, kind)
                seen = []
                ranges = agg.fuse_views_ranged(renderer, cams, probs, nparts=nparts, on_rows=lambda lo, hi: seen.append((lo, hi)))
                assert len(ranges) == nparts and ranges[0][0] == 0 and ranges[-1][1] == P
                assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:])) and all(lo % 64 == 0 for lo, _ in ranges)
                assert seen == [r for r in ranges if r[1] > r[0]]
                assert np.array_equal(agg.get_raw(), want), (C, kind, nparts)
                assert np.array_equal(agg.get(), whole.get())
    # a coarse mesh -- medium triangles (float atomics, all with part 0) and large ones (tail waves, taken by the part their position
    # falls into): the same sums to 1e-5, every row touched once
    from helpers import assert_fused_close
    mesh, cams = small_scene(24, 13, 320, 240, views=5)
    P, C = len(mesh.faces), 19
    renderer = sm.render.triangles(mesh)
    probs = [synth.device_probs(320, 240, C, synth.probs_seed(9, k), 0.05, 0) for k in range(len(cams))]
    for kind in ("sum", "summax", "mul"):
        whole = sm.fusion.MeshAggregator(P, C, kind)
        whole.fuse_views(renderer, cams, probs)
        for nparts in (2, 5):
            agg = sm.fusion.MeshAggregator(P, C, kind)
            agg.fuse_views_ranged(renderer, cams, probs, nparts=nparts)
            assert_fused_close(agg.get(), whole.get(), rtol=1e-5)
            np.testing.assert_allclose(agg.get_raw(), whole.get_raw(), rtol=1e-5, atol=1e-5)
    mesh, cams = small_scene(170, 81, 320, 240, views=2)
    P = len(mesh.faces)
    renderer = sm.render.triangles(mesh)
    # a part out of order is refused, and so is a part of a job that is over
    import ctypes
    from semantic_meshes_amd import _lib
    lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
    agg = sm.fusion.MeshAggregator(P, 19)
    assert _lib.lib().smesh_fuse_views_continue(renderer._h, agg._h, 1, ctypes.byref(lo), ctypes.byref(hi)) == 1   # SMESH_ERR_INVALID


def test_fuse_views_ranged_where_rows_are_not_in_triangle_order(sm):
    """Texel renderers and re-ordered (shuffled) meshes: part 0 is the whole job, rows (0, P); the later parts are empty."""
    from semantic_meshes_amd import synth
    from semantic_meshes_amd.data import Mesh
    from helpers import small_scene
    mesh, cams = small_scene(200, 100, 160, 120, views=3)
    rng = np.random.default_rng(5)
    shuffled = Mesh(mesh.vertices, mesh.faces[rng.permutation(len(mesh.faces))])     # (the renderer re-orders such a mesh: DESIGN.md 2, NOTES/kernels_rounds1-4.md "Face order")
    for n, renderer in enumerate((sm.render.triangles(shuffled), sm.render.texels(mesh, cams, 0.5))):
        P, C = renderer.getPrimitivesNum(), 7
        probs = [synth.device_probs(160, 120, C, synth.probs_seed(4, k), 0.0, 0) for k in range(3)]
        whole = sm.fusion.MeshAggregator(P, C)
        whole.fuse_views(renderer, cams, probs)
        agg = sm.fusion.MeshAggregator(P, C)
        ranges = agg.fuse_views_ranged(renderer, cams, probs, nparts=3)
        assert ranges == [(0, P), (P, P), (P, P)], (n, ranges)
        assert np.array_equal(agg.get_raw(), whole.get_raw())


def test_native_allreduce_rows_world_one_is_an_identity(sm):
    """`smesh_allreduce_rows` through a one-rank RCCL communicator -- the exchange stream, the events both ways, Mul's float64 staging --
    leaves the sums what they were: Sum bit-equal to the job without an exchange, Mul equal in get() (the (hi, lo) pairs are re-split)."""
    from semantic_meshes_amd import comm as smcomm, distributed as smdist, synth
    from helpers import small_scene, assert_fused_close
    mesh, cams = small_scene(170, 81, 320, 240, views=6)     # (two-pixel triangles: no float atomics, the additions have one order)
    P, C = len(mesh.faces), 19
    renderer = sm.render.triangles(mesh)
    probs = [synth.device_probs(320, 240, C, synth.probs_seed(2, k), 0.05, 0) for k in range(len(cams))]
    c = smcomm.Communicator(0, 0, 1, smcomm.Communicator.unique_id())
    assert c.nranks() == (0, 1)
    for kind in ("sum", "mul"):
        whole = sm.fusion.MeshAggregator(P, C, kind)
        whole.fuse_views(renderer, cams, probs)
        agg = sm.fusion.MeshAggregator(P, C, kind)
        _, rows = smdist.fuse_views_sharded(renderer, agg, cams, lambda k: probs[k], comm=c, nparts=4, held=4)
        assert rows == (0, P)
        if kind == "sum":
            assert np.array_equal(agg.get_raw(), whole.get_raw())
            assert np.array_equal(agg.get(), whole.get())
        else:
            assert_fused_close(agg.get(), whole.get(), rtol=1e-6)
        # the aggregator goes on working after the exchange (the main stream has picked the exchange stream up)
        agg.fuse_views(renderer, cams[:2], probs[:2])
        whole.fuse_views(renderer, cams[:2], probs[:2])
        assert_fused_close(agg.get(), whole.get(), rtol=1e-6)
        c.allreduce(agg)
        assert_fused_close(agg.get(), whole.get(), rtol=1e-6)


def test_cfg3_geometry_eight_ranks_on_one_gpu(tmp_path, sm):
    """BASELINE cfg3's shape on the one GPU of the test box: `bench.py --gpus 8 --workload cfg2` launched exactly as the driver launches it,
    eight ranks sharing the GPU (SMESH_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device), each with the 1 M-triangle mesh at
    1920 x 1080 -- the held views' records, the 76 MB exchange in four row ranges and the N > 1 branch of the script at real size.
    SMESH_BENCH_DUMP: rank 0 writes a sample of its get() rows, compared here with a single-process fusion of the same 64 views."""
    import json
    from semantic_meshes_amd import synth
    port = _free_port()
    dump = os.path.join(tmp_path, "rows.npz")
    env = dict(os.environ, SMESH_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", SMESH_BENCH_DUMP=dump)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SMESH_EXCHANGE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "8", "--warmup", "2", "--workload", "cfg2", "--repeats", "7"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 8 and cfg["nranks"] == 8 and cfg["allreduce_bytes"] == 76000000
    assert cfg["exchange_parts"] == 4 and cfg["held_views"] == 8
    assert abs(cfg["compute_ms"] + cfg["exchange_exposed_ms"] - cfg["timed_region_ms"]) < 0.35 * cfg["timed_region_ms"] + 5.0
    # the same 64 views (rank r: views r * 10 + 2 .. r * 10 + 9 of an 80-view ring) fused by ONE process
    got = np.load(dump)
    rows, fused = got["rows"], got["fused"]
    cfgd = synth.CONFIGS["cfg2"]
    W, H, C = cfgd["width"], cfgd["height"], cfgd["classes"]
    mesh = synth.grid_mesh(cfgd["a"], cfgd["b"])
    renderer = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(len(mesh.faces), C)
    ring = max(cfgd["views"], 8 * 10)
    for r in range(8):
        ids = [r * 10 + i for i in range(2, 10)]
        agg.fuse_views(renderer, [synth.ring_camera(k, ring, W, H) for k in ids],
                       [synth.device_probs(W, H, C, synth.probs_seed(1, k), 0.0, 0) for k in ids])
    want = agg.get()[rows]
    from helpers import assert_fused_close
    assert (want.sum(axis=1) > 0.5).sum() > len(rows) // 4
    assert_fused_close(fused, want, rtol=1e-5)


GROUPED = r'''
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.environ["SMESH_ROOT"])
sys.path.insert(0, os.path.join(os.environ["SMESH_ROOT"], "tests"))
import semantic_meshes_amd as sm
from semantic_meshes_amd import comm as smcomm, synth
from helpers import small_scene, assert_fused_close
mesh, cams = small_scene(60, 30, 320, 240, views=6)
P, C = len(mesh.faces), 19
renderer = sm.render.triangles(mesh)
probs = [synth.device_probs(320, 240, C, synth.probs_seed(6, k), 0.05, 0) for k in range(len(cams))]
comms = smcomm.create_all([0, 0, 0])          # three "GPUs" of one process (the mock's ranks), all on device 0
assert [c.nranks() for c in comms] == [(0, 3), (1, 3), (2, 3)]
for kind in ("sum", "summax", "mul"):
    whole = sm.fusion.MeshAggregator(P, C, kind)
    whole.fuse_views(renderer, cams, probs)
    aggs = [sm.fusion.MeshAggregator(P, C, kind) for _ in comms]
    for r, a in enumerate(aggs):
        a.fuse_views(renderer, cams[2 * r:2 * r + 2], probs[2 * r:2 * r + 2])
    partial = aggs[0].get()
    smcomm.allreduce_all(comms, aggs)
    for a in aggs:
        assert_fused_close(a.get(), whole.get(), rtol=1e-5)
    assert np.abs(partial - whole.get()).max() > 1e-3          # (the partial sums of one rank are NOT the fusion: the reduction did something)
mock = ctypes.CDLL(os.environ["SMESH_RCCL_LIB"])
assert mock.mock_rccl_groups_with_work() == 3, mock.mock_rccl_groups_with_work()
print("grouped ok")
'''


def test_grouped_allreduce_of_one_process_driving_several_gpus(tmp_path, sm):
    """`smesh_allreduce` with n > 1 (`comm.create_all` + `comm.allreduce_all`: one process, several GPUs, the collectives grouped).
    RCCL refuses two ranks per device, so on the one-GPU box the library is pointed at tests/mock_rccl.hip (SMESH_RCCL_LIB), which
    keeps RCCL's group semantics: a grouped ncclAllReduce reaches its stream at ncclGroupEnd.  ADVICE r4 (high): the Mul epilogue
    (float64 image -> (hi, lo) pairs) was launched inside the group, i.e. ahead of the reduction, and every GPU kept its own partial
    sums without an error.  Three ranks, Sum / Summax / Mul: every aggregator ends with the fusion of all six views."""
    mock = os.path.join(ROOT, "tests", "libmock_rccl.so")
    src = os.path.join(ROOT, "tests", "mock_rccl.hip")
    if not os.path.exists(mock) or os.path.getmtime(mock) < os.path.getmtime(src):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", mock, src])
    script = os.path.join(tmp_path, "grouped.py")
    open(script, "w").write(GROUPED)
    env = dict(os.environ, SMESH_ROOT=ROOT, SMESH_RCCL_LIB=mock)
    out = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "grouped ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
