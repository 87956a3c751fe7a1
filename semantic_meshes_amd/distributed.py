"""Data-parallel fusion over the GPUs of one node (new functionality).

Views are independent and every aggregator is a sum in some domain (Sum/Summax: weighted sums, Mul:
log-domain sums; /root/reference/python/semantic_meshes/src/Fusion.cu:46-92), so the path shards with no
data-path collective: each rank fuses its own views into a private accumulator and ONE sum all-reduce of
the raw float32[P, row_stride] buffer precedes get().  One process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU-only test machines) is used as plumbing only.
"""
import numpy as np


def shard_views(num_views, rank, world_size, contiguous=True):
    """Indices of the views rank `rank` fuses.  `contiguous` gives each rank one block (BASELINE cfg3:
    200 views per GPU); otherwise views are dealt round-robin (view k -> rank k % world_size)."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    if not contiguous:
        return list(range(rank, num_views, world_size))
    base, extra = divmod(num_views, world_size)
    start = rank * base + min(rank, extra)
    return list(range(start, start + base + (1 if rank < extra else 0)))


def _dist():
    import torch.distributed as dist
    return dist


def allreduce_raw(aggregator, group=None, comm=None):
    """Sum the un-normalised accumulators of all ranks in place; afterwards every rank's get() returns the
    fusion of ALL views.

    `comm` (a `semantic_meshes_amd.comm.Communicator`): the native path -- `smesh_allreduce`, ONE ncclAllReduce enqueued on
    the library's own stream right behind the fusion kernels, no host synchronisation, no PyTorch.
    Otherwise `torch.distributed` is used as plumbing: works for any object with get_raw()/set_raw(); a HIP aggregator under
    the nccl backend is reduced in place in HBM (torch's stream is ordered after the library's and back by events)."""
    if comm is not None:
        return comm.allreduce(aggregator)
    dist = _dist()
    import os
    if not dist.is_available() or not dist.is_initialized():
        return aggregator
    if dist.get_world_size(group) == 1 and not os.environ.get("SMESH_FORCE_ALLREDUCE"):
        return aggregator
    import torch
    backend = dist.get_backend(group)
    if backend == "nccl" and hasattr(aggregator, "raw_device_array"):
        flat = aggregator.raw_device_array(padded=True)         # float32[P * row_stride] view of HBM
        t = torch.as_tensor(flat, device="cuda:%d" % flat.device)  # zero-copy via __cuda_array_interface__
        if t.data_ptr() != flat.ptr:
            raise RuntimeError("torch copied the accumulator instead of aliasing it")
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        # later library work (get(), more views) is ordered after the collective on the device, not by a host wait
        from . import _lib
        import ctypes
        _lib.check(_lib.lib().smesh_stream_wait(flat.device, ctypes.c_void_p(int(torch.cuda.current_stream(flat.device).cuda_stream))))
        return aggregator
    if hasattr(aggregator, "get_raw_rows"):
        return allreduce_rows_raw(aggregator, 0, aggregator.primitives, group)     # (host copies; Mul's (hi, lo) pairs as float64)
    raw = np.ascontiguousarray(aggregator.get_raw(), dtype=np.float32)
    t = torch.from_numpy(raw)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    aggregator.set_raw(raw)
    return aggregator


def allreduce_rows_raw(aggregator, row_lo, row_hi, group=None, comm=None):
    """`allreduce_raw` for the rows [row_lo, row_hi) only -- what `MeshAggregator.fuse_views_ranged` hands to `on_rows` in a sharded job.
    `comm`: `smesh_allreduce_rows`, one ncclAllReduce of the range on the library's exchange stream, beside the fusion of the next
    range.  Otherwise through `torch.distributed` with host copies of the range (gloo: CPU-only test machines, several ranks sharing one
    GPU) -- synchronous, same sums.  Mul aggregators exchange their (hi, lo) float32 pairs as float64 either way."""
    if comm is not None:
        return comm.allreduce_rows(aggregator, row_lo, row_hi)
    dist = _dist()
    if not dist.is_available() or not dist.is_initialized() or row_hi <= row_lo:
        return aggregator
    import os
    if dist.get_world_size(group) == 1 and not os.environ.get("SMESH_FORCE_ALLREDUCE"):
        return aggregator          # (as allreduce_raw: nothing to add, no round trip through the host)
    import torch
    if dist.get_backend(group) == "nccl" and hasattr(aggregator, "raw_device_array"):
        # in place in HBM on torch's stream (Mul: the pair folded into the float32 hi plane first -- the exact float64 exchange is
        # the native path's); the library's streams are ordered behind the collective by an event
        import ctypes
        from . import _lib
        flat = aggregator.raw_device_array(padded=True)
        S = flat.shape[0] // max(aggregator.primitives, 1)
        t = torch.as_tensor(flat, device="cuda:%d" % flat.device)
        if t.data_ptr() != flat.ptr:
            raise RuntimeError("torch copied the accumulator instead of aliasing it")
        dist.all_reduce(t[row_lo * S:row_hi * S], op=dist.ReduceOp.SUM, group=group)
        _lib.check(_lib.lib().smesh_stream_wait(flat.device, ctypes.c_void_p(int(torch.cuda.current_stream(flat.device).cuda_stream))))
        return aggregator
    if getattr(aggregator, "kind", None) == "Mul":
        v = aggregator.get_raw_rows(row_lo, row_hi, 0).astype(np.float64) + aggregator.get_raw_rows(row_lo, row_hi, 1)
        t = torch.from_numpy(v)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        with np.errstate(invalid="ignore"):
            hi = v.astype(np.float32)
            lo = np.where(np.isfinite(hi), v - hi, 0.0).astype(np.float32)
        aggregator.set_raw_rows(row_lo, hi, 0)
        aggregator.set_raw_rows(row_lo, lo, 1)
        return aggregator
    raw = aggregator.get_raw_rows(row_lo, row_hi, 0)
    t = torch.from_numpy(raw)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    aggregator.set_raw_rows(row_lo, raw, 0)
    return aggregator


def part_rows(primitives, part, nparts):
    """Accumulator rows [lo, hi) that part `part` of `nparts` of a job cut by row range leaves final (`smesh_fuse_part_rows`: whole
    64-row blocks).  A function of its arguments alone -- every rank of a sharded job derives the same ranges from it."""
    P, part, nparts = int(primitives), int(part), int(nparts)
    blocks = (P + 63) // 64
    return min(P, 64 * (blocks * part // nparts)), min(P, 64 * (blocks * (part + 1) // nparts))


def owned_rows(primitives, rank, world_size):
    """Rows [lo, hi) of the accumulator that rank `rank` owns after a reduce-scatter: q = floor(P / world / 4) * 4 rows per
    rank (16-byte aligned slices), the remainder counted into the last rank's range (`smesh_reduce_scatter`)."""
    q = (int(primitives) // int(world_size)) & ~3
    return q * rank, (int(primitives) if rank == world_size - 1 else q * (rank + 1))


def reduce_scatter_raw(aggregator, group=None, comm=None):
    """Opt-in exchange (SMESH_EXCHANGE=reduce_scatter): afterwards this rank's rows `owned_rows(P, rank, world)` hold the sum over
    all ranks; returns that `(row_lo, row_hi)`.  Native: one in-place ncclReduceScatter on the library's stream.  Through
    torch.distributed: `reduce_scatter_tensor` in place under nccl; under gloo (CPU test machines, several ranks sharing one GPU)
    the all-reduce of `allreduce_raw` stands in -- same rows, same values."""
    if comm is not None:
        return comm.reduce_scatter(aggregator)
    dist = _dist()
    P = aggregator.primitives
    if not dist.is_available() or not dist.is_initialized():
        return 0, P
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = owned_rows(P, rank, world)
    if dist.get_backend(group) == "nccl" and hasattr(aggregator, "raw_device_array") and world > 1:
        import ctypes
        import torch
        from . import _lib
        flat = aggregator.raw_device_array(padded=True)
        S = flat.shape[0] // max(P, 1)
        t = torch.as_tensor(flat, device="cuda:%d" % flat.device)
        q = (P // world) & ~3
        if q:
            dist.reduce_scatter_tensor(t[lo * S:(lo + q) * S], t[:q * world * S], op=dist.ReduceOp.SUM, group=group)
        if q * world < P:
            dist.all_reduce(t[q * world * S:], op=dist.ReduceOp.SUM, group=group)
        _lib.check(_lib.lib().smesh_stream_wait(flat.device, ctypes.c_void_p(int(torch.cuda.current_stream(flat.device).cuda_stream))))
        return lo, hi
    allreduce_raw(aggregator, group)
    return lo, hi


def fuse_views_sharded(renderer, aggregator, cameras, probs_of_view, group=None, contiguous=True, batch=8, comm=None,
                       exchange="allreduce", nparts=4, held=24):
    """Fuse this rank's share of `cameras` and exchange.  `probs_of_view(k)` returns the (W,H,C) class-probability image of view k
    (host or device).  `comm`: native communicator (see allreduce_raw).  Returns `(aggregator, (row_lo, row_hi))`: the rows of
    THIS rank's accumulator that hold the fusion of all views afterwards.  (Until round 3 the function returned the aggregator
    alone -- or the row range alone for a reduce-scatter -- and read SMESH_EXCHANGE; callers of that contract unpack the pair now.)

    `exchange` (an explicit argument, never the environment):
      "allreduce" (default, what north_star names) -- every rank ends with the whole fusion, rows (0, P).  With `nparts` > 1 the last
         `held` (at most 32) views of the rank are fused by accumulator row range (`MeshAggregator.fuse_views_ranged`) and the
         all-reduce of every finished range is queued on the exchange stream while the next range is fused: the same ONE sum over the
         same bytes, in `nparts` pieces, of which only the last is exposed.  `nparts` = 1: one all-reduce after the last view.
      "reduce_scatter" -- one in-place reduce-scatter after the last view; rank r owns `owned_rows(P, r, world)`, fetch them with
         `aggregator.get_rows(row_lo, row_hi)` (everything else is refused until `reset()`)."""
    if exchange not in ("allreduce", "reduce_scatter"):
        raise ValueError("exchange must be 'allreduce' or 'reduce_scatter'")
    if comm is not None:
        rank, world = comm.rank, comm.world
    else:
        dist = _dist()
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
        else:
            rank, world = 0, 1
    mine = list(shard_views(len(cameras), rank, world, contiguous))
    # Every rank must issue the SAME sequence of collectives whatever its share of the views is (a rank without views, a rank whose
    # library call fell back to the unranged job): whether the exchange is cut, and into which row ranges, is a function of
    # (primitives, nparts) alone -- `part_rows` -- never of this rank's views.
    ranged = exchange == "allreduce" and int(nparts) > 1 and hasattr(aggregator, "fuse_views_ranged")
    tail = mine[max(0, len(mine) - max(1, min(int(held), 32))):] if ranged else []
    head = mine[:len(mine) - len(tail)]
    if hasattr(aggregator, "fuse_views"):
        # eight views per call: the library shares rasteriser and fusion launches between them
        for b in range(0, len(head), batch):
            ks = head[b:b + batch]
            aggregator.fuse_views(renderer, [cameras[k] for k in ks], [probs_of_view(k) for k in ks])
    else:
        for k in head:
            idx, _ = renderer.render(cameras[k])
            aggregator.add(idx, probs_of_view(k))
    if exchange == "reduce_scatter":
        return aggregator, tuple(reduce_scatter_raw(aggregator, group, comm))
    if ranged:
        canon = [r for r in (part_rows(aggregator.primitives, p, int(nparts)) for p in range(int(nparts))) if r[1] > r[0]]
        sent = []

        def on_rows(lo, hi):
            # a finished range is exchanged at once if it is the next one of the canonical sequence; anything else (the library
            # fused everything with part 0: host images, a re-ordered mesh, a texel renderer ...) waits for the end of the call
            if len(sent) < len(canon) and (lo, hi) == canon[len(sent)]:
                sent.append((lo, hi))
                allreduce_rows_raw(aggregator, lo, hi, group, comm)

        if tail:
            images = [probs_of_view(k) for k in tail]      # (kept alive until the last part has been queued)
            aggregator.fuse_views_ranged(renderer, [cameras[k] for k in tail], images, nparts=int(nparts), on_rows=on_rows)
            del images
        for lo, hi in canon[len(sent):]:                    # (a rank without views; ranges the call above did not report)
            allreduce_rows_raw(aggregator, lo, hi, group, comm)
    else:
        allreduce_raw(aggregator, group, comm)
    return aggregator, (0, aggregator.primitives)
