"""GPU tests of the exchange step and of the stream contract with other frameworks.

* native `smesh_allreduce` (RCCL on the library's stream, no PyTorch): world size 1 here (the GPU box has one device;
  two devices when present), `get()` equal to the single-rank job;
* the torch.distributed `nccl` in-place path with a forced world-size-1 all-reduce: aliasing, no copy;
* device inputs produced by torch on ITS stream are ordered before the library's kernels (the library's streams
  are non-blocking);
* `__cuda_array_interface__` of a render() result waits for the rasteriser.
"""
import ctypes
import os
import socket

import numpy as np
import pytest

from helpers import assert_fused_close, random_probs, small_scene

pytestmark = pytest.mark.gpu


def _fuse_all(sm, mesh, cams, probs, C, kind="sum", device=0):
    r = sm.render.triangles(mesh, device=device)
    agg = sm.fusion.MeshAggregator(len(mesh.faces), C, kind, device=device)
    for cam, p in zip(cams, probs):
        agg.fuse_view(r, cam, p)
    return r, agg


def test_native_allreduce_world_one_is_identity_and_async(sm):
    from semantic_meshes_amd import comm, _lib
    mesh, cams = small_scene(40, 20, 160, 120, views=4)
    C = 19
    rng = np.random.default_rng(5)
    probs = [random_probs(rng, 160, 120, C) for _ in cams]
    _, whole = _fuse_all(sm, mesh, cams, probs, C)
    want_raw, want = whole.get_raw(), whole.get()
    c = comm.Communicator(0, 0, 1, comm.Communicator.unique_id())
    r, agg = _fuse_all(sm, mesh, cams, probs, C)
    c.allreduce(agg)                      # enqueued behind the fusion kernels, no host wait in between
    np.testing.assert_array_equal(agg.get_raw(), want_raw)
    np.testing.assert_array_equal(agg.get(), want)
    assert c.reduce_scalars([1.5, -2.0], "max") == [1.5, -2.0]
    assert c.reduce_scalars([3.0], "sum") == [3.0]
    c.barrier()
    # from_env with the launcher's variables, world size 1
    env = dict(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29431")
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        c2 = comm.Communicator.from_env()
        c2.allreduce(agg)
        np.testing.assert_array_equal(agg.get(), want)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    # errors: a communicator of another device / NULL handles are rejected, not crashed on
    with pytest.raises(ValueError):
        _lib.check(_lib.lib().smesh_allreduce(None, None, 1))


def test_native_allreduce_all_devices_of_this_box(sm):
    """smesh_comm_create_all over every visible GPU (1 on the round's box, 8 on a node): views dealt round-robin,
    one grouped all-reduce, every device's get() equals the single-device fusion within the float tolerance."""
    from semantic_meshes_amd import comm, _lib
    ndev = _lib.device_count()
    mesh, cams = small_scene(36, 18, 128, 96, views=max(4, 2 * ndev))
    C = 7
    rng = np.random.default_rng(11)
    probs = [random_probs(rng, 128, 96, C) for _ in cams]
    _, whole = _fuse_all(sm, mesh, cams, probs, C)
    want = whole.get()
    comms = comm.create_all(list(range(ndev)))
    aggs, keep = [], []
    for d in range(ndev):
        r, a = _fuse_all(sm, mesh, cams[d::ndev], probs[d::ndev], C, device=d)
        aggs.append(a)
        keep.append(r)
    comm.allreduce_all(comms, aggs)
    for a in aggs:
        got = a.get()
        if ndev == 1:
            np.testing.assert_array_equal(got, want)
        else:
            assert_fused_close(got, want)


def _torch():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.fail("torch sees no GPU on a gpu-marked run")
    return torch


def test_torch_nccl_in_place_allreduce_world_one(sm):
    """distributed.allreduce_raw under backend nccl: torch aliases the padded accumulator (no copy), the all-reduce of a
    world of one leaves it bit-identical, and the library's later work is ordered after torch's stream."""
    torch = _torch()
    import torch.distributed as dist
    from semantic_meshes_amd import distributed as smdist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        mesh, cams = small_scene(40, 20, 160, 120, views=3)
        C = 19
        rng = np.random.default_rng(6)
        probs = [random_probs(rng, 160, 120, C) for _ in cams]
        _, agg = _fuse_all(sm, mesh, cams, probs, C)
        want_raw = agg.get_raw()
        flat = agg.raw_device_array(padded=True)
        t = torch.as_tensor(flat, device="cuda:0")
        assert t.data_ptr() == flat.ptr and t.numel() == flat.size
        os.environ["SMESH_FORCE_ALLREDUCE"] = "1"
        try:
            smdist.allreduce_raw(agg)
        finally:
            os.environ.pop("SMESH_FORCE_ALLREDUCE", None)
        np.testing.assert_array_equal(agg.get_raw(), want_raw)
        # and the sharded helper end to end (world 1: every view is this rank's)
        r = sm.render.triangles(mesh)
        agg2 = sm.fusion.MeshAggregator(len(mesh.faces), C)
        smdist.fuse_views_sharded(r, agg2, cams, lambda k: probs[k])
        np.testing.assert_array_equal(agg2.get_raw(), want_raw)
    finally:
        dist.destroy_process_group()


def test_torch_produced_probs_are_ordered_before_the_fusion(sm):
    """probs = net(x) on torch's stream; add(idx, probs) right away.  The library's kernels must see the finished tensor:
    a long chain of torch kernels writes the probabilities last.  Reference result: the same views fused from host
    arrays (the kernels are deterministic, so the accumulators must be bit-equal)."""
    torch = _torch()
    mesh, cams = small_scene(60, 30, 640, 480, views=2)
    W, H, C = 640, 480, 19
    P = len(mesh.faces)
    rng = np.random.default_rng(8)
    base = random_probs(rng, W, H, C, 0.0)
    r = sm.render.triangles(mesh)
    ref = sm.fusion.MeshAggregator(P, C)
    for cam in cams:
        idx, _ = r.render(cam)
        ref.add(idx, base)
    want = ref.get_raw()
    assert np.abs(want).sum() > 0
    for trial in range(3):
        agg = sm.fusion.MeshAggregator(P, C)
        for cam in cams:
            idx, _ = r.render(cam)
            t = torch.zeros((W, H, C), device="cuda:0")
            big = torch.ones((4096, 4096), device="cuda:0")
            for _ in range(6):                       # keep torch's stream busy for a few milliseconds ...
                big = big @ big * 1e-4
            t += torch.from_numpy(base).to("cuda:0") * (big[0, 0] * 0 + 1)    # ... and only then write the probabilities
            agg.add(idx, t)                          # no torch.cuda.synchronize() in between
            del t                                    # add() does not retain its inputs: torch may reuse the block at once
        np.testing.assert_array_equal(agg.get_raw(), want)


def test_cuda_array_interface_waits_for_the_render(sm, oracle):
    torch = _torch()
    from semantic_meshes_amd import synth
    mesh, cams, _ = synth.scene("cfg2")
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    for k in (3, 50):
        idx, depth = r.render(cams[k])
        t = torch.as_tensor(idx, device="cuda:0").clone()     # consumed on torch's stream straight away
        oidx, _ = o.render(cams[k])
        np.testing.assert_array_equal(t.cpu().numpy().view(np.uint32), oidx)


def test_profile_counts_launches_and_views(sm):
    """smesh_profile_read_ex: launches / views of the dominant kernel inside the timed regions (bench.py's roofline divisor)."""
    from semantic_meshes_amd import _lib, synth
    mesh, cams = small_scene(60, 30, 320, 240, views=11)
    C = 19
    P = len(mesh.faces)
    probs = [synth.device_probs(320, 240, C, 40 + k) for k in range(len(cams))]
    r = sm.render.triangles(mesh)
    agg = sm.fusion.MeshAggregator(P, C)
    agg.defer = False          # (the library's counters per LIBRARY call: the Python layer must not group the fuse_view calls below)
    L = _lib.lib()

    def read():
        ms, reg, n, v = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(L.smesh_profile_read_ex(0, _lib.PROF_FUSE_SCATTER, ctypes.byref(ms), ctypes.byref(reg), ctypes.byref(n), ctypes.byref(v)))
        return ms.value, reg.value, n.value, v.value

    agg.fuse_views(r, cams[:8], probs[:8])       # warm
    _lib.check(L.smesh_profile_reset(0))
    _lib.check(L.smesh_profile_sample_every(0, 1))
    _lib.check(L.smesh_profile_enable(0, 1 << _lib.PROF_FUSE_SCATTER))
    try:
        agg.fuse_views(r, cams[:8], probs[:8])   # one region: ONE launch of 8 views (class counts up to 24)
        ms, reg, n, v = read()
        assert (reg, n, v) == (1, 1, 8) and ms > 0
        agg.fuse_views(r, cams[8:11], probs[8:11])   # group of 3: a two-view launch and a single
        ms, reg, n, v = read()
        assert (reg, n, v) == (2, 3, 11)
        agg.fuse_view(r, cams[0], probs[0])
        assert read()[1:] == (3, 4, 12)
        agg.fuse_views(r, cams[:7], probs[:7])   # 7 = 4 + 2 + 1
        assert read()[1:] == (4, 7, 19)
        _lib.check(L.smesh_profile_sample_every(0, 2))   # every second region: counters follow the TIMED regions only
        _lib.check(L.smesh_profile_reset(0))
        for _ in range(4):
            agg.fuse_view(r, cams[1], probs[1])
        assert read()[1:] == (2, 2, 2)
    finally:
        _lib.check(L.smesh_profile_enable(0, 0))
        _lib.check(L.smesh_profile_sample_every(0, 1))


def test_render_can_return_the_reference_s_dltensor_capsules(sm, oracle):
    """python/semantic_meshes/include/Renderer.h:37-38: render() returns PyCapsules named "dltensor".  Behind a switch here;
    a raw-capsule consumer (torch.utils.dlpack.from_dlpack stands in for tf.experimental.dlpack.from_dlpack,
    eval_scannet.py:211-212) takes them, and add() takes an unconsumed one as python/scripts/colorize_cityscapes_mesh.py:65-67 does."""
    torch = _torch()
    import torch.utils.dlpack
    mesh, cams = small_scene(120, 60, 320, 240, views=3)
    P, C = len(mesh.faces), 19
    rng = np.random.default_rng(3)
    r = sm.render.triangles(mesh)
    o = oracle.OracleRenderer(mesh.vertices, mesh.faces)
    agg, oagg = sm.fusion.MeshAggregator(P, C), oracle.OracleAggregator(P, C)
    for cam in cams:
        idx_c, depth_c = r.render(cam, capsules=True)
        assert type(idx_c).__name__ == "PyCapsule" and type(depth_c).__name__ == "PyCapsule"
        depth_t = torch.utils.dlpack.from_dlpack(depth_c)                 # consumed by another framework
        oidx, odepth = o.render(cam)
        assert depth_t.shape == cam.resolution and depth_t.is_cuda
        np.testing.assert_array_equal(depth_t.cpu().numpy().view(np.uint32), odepth.view(np.uint32))
        probs = random_probs(rng, *cam.resolution, C)
        agg.add(idx_c, probs)                                             # the unconsumed capsule goes straight into add()
        assert sm._lib.last_fuse_kernel() == "k_fuse_tri"
        oagg.add(oidx, probs)
        with pytest.raises(ValueError):
            agg.add(idx_c, probs)                                         # a capsule is consumed once
    assert_fused_close(agg.get(), oagg.get(), rtol=1e-5)
    # ... and a consumed index capsule comes back as the other framework's tensor: recognised by content
    idx_c, _ = r.render(cams[0], capsules=True)
    idx_t = torch.utils.dlpack.from_dlpack(idx_c)
    agg.add(idx_t, to_dev(sm, random_probs(rng, *cams[0].resolution, C)))
    assert sm._lib.last_fuse_kernel() == "k_fuse_tri"


def to_dev(sm, a):
    from semantic_meshes_amd.device import to_device
    return to_device(a)


@pytest.mark.parametrize("variant", ["native", "torch"])
def test_bench_under_the_drivers_launcher_world_one(variant):
    """bench.py the way the driver starts it for N > 1 (`python -m torch.distributed.run ... bench.py --gpus N`), with one rank:
    RANK / WORLD_SIZE / MASTER_* from the environment, the unique id over the bootstrap socket, communicator creation, the
    all-reduce inside the timed region, the barrier and the max-over-ranks clock -- native RCCL path and torch.distributed variant."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SMESH_ALLREDUCE=variant)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "16", "--warmup", "8",
                          "--workload", "cfg1", "--no-host-path"], env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 16 and out["value"] > 100
    assert out["config"]["allreduce"].startswith("native" if variant == "native" else "torch.distributed")
    assert out["config"]["host_syncs_in_timed_region"] == (1 if variant == "native" else 7)
    # launched (RANK set): the rank's 16 views are all held and fused by row range -- two groups of eight views x four row ranges, each
    # range's all-reduce on the exchange stream (an identity with one rank); every fusion region is bracketed
    cfg = out["config"]
    assert cfg["exchange_parts"] == 4 and cfg["held_views"] == 16 and cfg["nranks"] == 1
    rr = cfg["exchange_row_ranges"]
    assert rr[0][0] == 0 and rr[-1][1] == 10000 and all(a[1] == b[0] for a, b in zip(rr, rr[1:]))
    assert cfg["compute_ms"] > 0 and cfg["exchange_exposed_ms"] >= 0 and cfg["exchange_ms"] >= 0
    assert cfg["compute_ms"] + cfg["exchange_exposed_ms"] <= cfg["timed_region_ms"] * 1.05
    assert out["roofline"]["regions_timed"] == 0 and out["roofline"]["note"]      # (no event pairs around the cut fusion launches: see the note)


@pytest.mark.parametrize("C,res", [(19, (320, 240)), (19, (333, 257)), (5, (37, 29)), (40, (320, 240)), (150, (320, 240))])
def test_permuted_device_probs_take_the_render_records_path(sm, oracle, C, res):
    """A network's (H,W,C) output seen as (W,H,C) -- torch's `permute(1, 0, 2)`: a strided view in device memory, no copy -- handed
    to add() with the untouched output of render(): the class vectors are gathered into the aggregator's scratch on the library's
    stream and the view takes the triangle-order kernels on the rasteriser's records, bit-equal to the dense image."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("torch sees no GPU")
    mesh, cams = small_scene(120, 60, res[0], res[1], views=3)   # (images that are no multiple of the gather's 32 x 8 tiles too)
    P = len(mesh.faces)
    rng = np.random.default_rng(C)
    r = sm.render.triangles(mesh)
    a, b = sm.fusion.MeshAggregator(P, C), sm.fusion.MeshAggregator(P, C)
    oagg = oracle.OracleAggregator(P, C)
    for k, cam in enumerate(cams):
        W, H = cam.resolution
        dense = random_probs(rng, W, H, C)                                   # (W,H,C)
        hwc = torch.from_numpy(np.ascontiguousarray(dense.transpose(1, 0, 2))).cuda()   # what the network produced: (H,W,C)
        view = hwc.permute(1, 0, 2)                                          # (W,H,C), strides (C, W*C, 1)
        assert not view.is_contiguous()
        idx, _ = r.render(cam)
        a.add(idx, view)
        assert sm._lib.last_add_path() == ("scatter" if os.environ.get("SMESH_FUSE") == "strip" else "render-records")
        b.fuse_view(r, cam, dense)
        oagg.add(np.asarray(idx), dense)
    sm._lib.synchronize(0)
    if os.environ.get("SMESH_FUSE") != "strip":
        np.testing.assert_array_equal(a.get_raw().view(np.uint32), b.get_raw().view(np.uint32))
    assert_fused_close(a.get(), oagg.get())


def test_reference_package_name_returns_capsules_by_default(sm):
    """`import semantic_meshes` (the reference's name): render() returns the reference's "dltensor" capsules without
    any switch, and the reference's loop (colorize_cityscapes_mesh.py:65-67: capsule straight back into add()) takes the
    triangle-order kernels."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import semantic_meshes
from semantic_meshes_amd import _lib, synth
from helpers import small_scene
mesh, cams = small_scene(40, 20, 160, 120, views=2)
r = semantic_meshes.render.triangles(mesh)
agg = semantic_meshes.fusion.MeshAggregator(primitives=r.getPrimitivesNum(), classes=5)
ref = semantic_meshes.fusion.MeshAggregator(primitives=r.getPrimitivesNum(), classes=5)
for k, cam in enumerate(cams):
    idx, depth = r.render(cam)
    assert type(idx).__name__ == "PyCapsule" and type(depth).__name__ == "PyCapsule"
    probs = np.asarray(synth.device_probs(160, 120, 5, 11 + k))
    agg.add(idx, probs)
    assert _lib.last_fuse_kernel().startswith("k_fuse_tri")
    ref.fuse_view(r, cam, probs)
assert np.array_equal(agg.get(), ref.get())
# ... and that is a property of the renderers made under the reference's name, not a switch on the shared implementation module
# (ADVICE r3): the same process keeps getting DeviceArrays from semantic_meshes_amd, whatever the import order
import semantic_meshes_amd
r2 = semantic_meshes_amd.render.triangles(mesh)
idx2, depth2 = r2.render(cams[0])
assert isinstance(idx2, semantic_meshes_amd.device.DeviceArray) and np.asarray(idx2).shape == (160, 120)
assert semantic_meshes_amd.render.RETURN_CAPSULES is False
idx3, _ = semantic_meshes.render.triangles(mesh, capsules=False).render(cams[0])
assert isinstance(idx3, semantic_meshes_amd.device.DeviceArray)
print("capsules ok")
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "SMESH_RENDER_CAPSULES"}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "capsules ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_device_block_cache_is_bounded_and_trimmable(sm):
    """the cache of still-mapped device blocks (csrc/context.cpp) is capped PER DEVICE at a sixteenth of its memory
    (at most 8 GiB) -- a block larger than that is really freed, so another allocator of the process finds the memory -- and
    `device.trim()` hands back the rest on request; a small block is recycled at the same address."""
    from semantic_meshes_amd import device as smdev
    smdev.trim()
    small = smdev.DeviceBuffer(3 << 20)
    p0 = small.ptr
    del small
    again = smdev.DeviceBuffer(3 << 20)
    assert again.ptr == p0                        # recycled, not re-mapped
    del again
    held = smdev.trim(0)
    assert held >= (3 << 20)
    assert smdev.trim(0) == 0                     # nothing left to give back
    big = smdev.DeviceBuffer(9 << 30)             # beyond the cap: must not stay mapped after the free
    del big
    assert smdev.trim(0) < (9 << 30)
