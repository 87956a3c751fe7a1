"""world_size-2 and -8 gloo tests of the N > 1 path: view sharding + one sum all-reduce of the raw accumulator
equals fusing every view on one rank.  Runs on CPU; the per-rank compute is the oracle
aggregator (test infrastructure) because the product has no CPU path."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["SMESH_ROOT"])
sys.path.insert(0, os.path.join(os.environ["SMESH_ROOT"], "tests"))
import torch.distributed as dist
from oracle import oracle
from semantic_meshes_amd import distributed as smdist, synth
from helpers import small_scene

dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
mesh, cams = small_scene(30, 16, 96, 72, views=5)
P, C = len(mesh.faces), 6
oracle.set_threads(1)
oracle.set_accum_double(True)   # the float32 all-reduce is what is under test, not float32 summation order
r = oracle.OracleRenderer(mesh.vertices, mesh.faces)

def probs_of_view(k):
    W, H = cams[k].resolution
    return oracle.synth_probs(W * H, C, synth.probs_seed(3, k), 0.1).reshape(W, H, C)

for kind in ("sum", "summax", "mul"):
    agg = oracle.OracleAggregator(P, C, kind, 0.5)
    smdist.fuse_views_sharded(r, agg, cams, probs_of_view, contiguous=(kind != "summax"))
    whole = oracle.OracleAggregator(P, C, kind, 0.5)
    for k in range(len(cams)):
        whole.add(r.render(cams[k])[0], probs_of_view(k))
    # Mul sums float32 LOG-probabilities of magnitude ~1e2: rounding the partial sums to float32 for the
    # all-reduce costs ~1e-5 absolute in the log domain, i.e. ~1e-5..1e-4 relative after exp()
    rtol = 1e-5     # (Mul: the exchange carries (hi, lo) pairs as float64)
    np.testing.assert_allclose(agg.get(), whole.get(), rtol=rtol, atol=2e-7)
    np.testing.assert_allclose(agg.get_raw(), whole.get_raw(), rtol=1e-5, atol=1e-6)
    assert (whole.get().sum(axis=1) > 0.5).sum() > P // 3
    # the opt-in exchange: every rank ends up owning a slice of rows (here through gloo, where the all-reduce stands in)
    agg2 = oracle.OracleAggregator(P, C, kind, 0.5)
    _, (lo, hi) = smdist.fuse_views_sharded(r, agg2, cams, probs_of_view, exchange="reduce_scatter")
    assert (lo, hi) == smdist.owned_rows(P, rank, world) and lo % 4 == 0
    np.testing.assert_allclose(agg2.get_rows(lo, hi), whole.get()[lo:hi], rtol=rtol, atol=2e-7)
    spans = [smdist.owned_rows(P, k, world) for k in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == P and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
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


import pytest


@pytest.mark.parametrize("world", [2, 8])
def test_ranks_gloo_equal_single_rank(tmp_path, world):
    """Two ranks, and cfg3's eight: with five views, three of the eight ranks have no view of their own and still take part in the exchange."""
    script = os.path.join(tmp_path, "worker.py")
    open(script, "w").write(WORKER)
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SMESH_ROOT=ROOT, OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, script], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            p.kill()
            out, _ = p.communicate()
        outs.append(out.decode(errors="replace"))
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (rank, out[-3000:])
        assert "rank %d ok" % rank in out
