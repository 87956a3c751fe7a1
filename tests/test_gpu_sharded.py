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

def probs_of_view(k):
    W, H = cams[k].resolution
    return synth.device_probs(W, H, C, synth.probs_seed(3, k), 0.05, 0)

out = {}
for kind in ("sum", "summax", "mul"):
    whole = sm.fusion.MeshAggregator(P, C, kind)
    whole.fuse_views(renderer, cams, [probs_of_view(k) for k in range(len(cams))])
    want, want_raw = whole.get(), whole.get_raw()
    agg = sm.fusion.MeshAggregator(P, C, kind)
    smdist.fuse_views_sharded(renderer, agg, cams, probs_of_view, contiguous=(kind != "summax"))
    kernel = _lib.lib().smesh_last_fuse_kernel().decode()
    assert kernel.startswith("k_fuse_tri"), kernel            # the HIP triangle-order path ran on this rank's shard
    got = agg.get()
    # partial float32 sums added once instead of eight terms in order: 1e-5 (Mul: log-domain partial sums rounded to float32)
    assert_fused_close(got, want, rtol=2e-4 if kind == "mul" else 1e-5)
    np.testing.assert_allclose(agg.get_raw(), want_raw, rtol=1e-5, atol=1e-4 if kind == "mul" else 1e-5)   # (Mul: log-domain rows of magnitude 10-20, up to eight float32 partial sums)
    assert (want.sum(axis=1) > 0.5).sum() > P // 3
    # opt-in exchange: this rank normalises its own slice of rows
    agg2 = sm.fusion.MeshAggregator(P, C, kind)
    lo, hi = smdist.fuse_views_sharded(renderer, agg2, cams, probs_of_view, exchange="reduce_scatter")
    assert (lo, hi) == smdist.owned_rows(P, rank, world)
    assert_fused_close(agg2.get_rows(lo, hi), want[lo:hi], rtol=2e-4 if kind == "mul" else 1e-5)
    out[kind] = got
np.savez(os.environ["SMESH_OUT"], **out)
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_on_one_gpu_hip_aggregators_equal_the_single_process_job(tmp_path, sm, oracle, world):
    """Two ranks, and cfg3's eight (one view per rank of the eight-view scene), sharing the test box's GPU."""
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMESH_ROOT=ROOT, SMESH_OUT=os.path.join(tmp_path, "rank%d.npz" % rank), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-3000:])
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
            assert_fused_close(r0[kind], o_a.get(), rtol=2e-4 if kind == "mul" else 1e-5)
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
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4", "--workload", "cfg1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    cfg = d["config"]
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["warmup"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert abs(d["value"] - 2 * 12 / (d["ms_per_step"] * 12 * 1e-3)) < 0.02 * d["value"]          # whole-job views / max-over-ranks time
    assert cfg["nranks"] == 2 and cfg["exchange"] == exchange and cfg["allreduce_bytes"] == 4 * 10000 * 5
    assert cfg["compute_ms"] > 0 and cfg["exchange_ms"] > 0 and cfg["exchange_ms"] >= cfg["exchange_ms_fastest_rank"]
    assert cfg["compute_ms"] + cfg["exchange_ms_fastest_rank"] <= cfg["timed_region_ms"] * 1.05
    assert "cpu_baseline" not in d and d["roofline"]["frac"] > 0
