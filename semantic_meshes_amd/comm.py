"""Native communicator for the sharded path: one RCCL sum all-reduce of the raw accumulators (include/smesh.h,
`smesh_comm_*`, `smesh_allreduce`; new functionality, SURVEY.md 8e -- the reference is single-GPU).

No PyTorch here: the 128-byte RCCL unique id travels from rank 0 to the other ranks over a plain TCP socket
(`exchange_id`), everything else happens inside libsmesh_hip.so on the library's own HIP stream.

    comm = Communicator.from_env(device)        # RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, one process per GPU
    ... agg.fuse_views(...) on every rank's shard of the views ...
    comm.allreduce(agg)                          # asynchronous, right behind the fusion kernels
    fused = agg.get()
"""
import ctypes
import os
import socket
import struct
import time

from . import _lib

ID_BYTES = 128
_MAGIC = b"SMESHID1"


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the bootstrap connection")
        buf += chunk
    return buf


def exchange_id(payload, rank, world, addr="127.0.0.1", port=29517, timeout=120.0, span=8):
    """Rank 0 hands `payload` (bytes) to ranks 1 .. world-1; returns the payload on every rank.

    Rank 0 listens on the first free port of [port, port + span); the others try those ports in turn until one
    answers with the handshake of this job (magic + world size), so a stray listener on one of them is skipped.
    Afterwards every rank has the payload and rank 0 has seen all `world - 1` peers."""
    if world <= 1:
        return payload
    deadline = time.time() + timeout
    if rank == 0:
        srv = None
        for p in range(port, port + span):
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind(("127.0.0.1" if addr in ("127.0.0.1", "localhost") else "", p))
                break
            except OSError:
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError("bootstrap: no free port in [%d, %d)" % (port, port + span))
        srv.listen(world)
        srv.settimeout(1.0)
        served = set()
        try:
            while len(served) < world - 1:
                if time.time() > deadline:
                    raise TimeoutError("bootstrap: only %d of %d peers connected" % (len(served), world - 1))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, len(_MAGIC) + 8)
                        if hello[:len(_MAGIC)] != _MAGIC:
                            continue
                        peer_world, peer_rank = struct.unpack("<ii", hello[len(_MAGIC):])
                        if peer_world != world or not (0 < peer_rank < world):
                            continue
                        conn.sendall(_MAGIC + struct.pack("<i", len(payload)) + payload)
                        served.add(peer_rank)
                    except (OSError, ConnectionError):
                        continue
        finally:
            srv.close()
        return payload
    while True:
        for p in range(port, port + span):
            try:
                with socket.create_connection((addr, p), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(_MAGIC + struct.pack("<ii", world, rank))
                    head = _recv_exact(conn, len(_MAGIC) + 4)
                    if head[:len(_MAGIC)] != _MAGIC:
                        continue
                    (n,) = struct.unpack("<i", head[len(_MAGIC):])
                    return _recv_exact(conn, n)
            except (OSError, ConnectionError):
                continue
        if time.time() > deadline:
            raise TimeoutError("bootstrap: rank 0 did not answer on %s:%d..%d" % (addr, port, port + span - 1))
        time.sleep(0.05)


class Communicator:
    """One rank of an RCCL communicator bound to one GPU (`smesh_comm_t`)."""

    def __init__(self, device, rank, world, unique_id):
        self.device, self.rank, self.world = int(device), int(rank), int(world)
        if len(unique_id) != ID_BYTES:
            raise ValueError("unique id must be %d bytes" % ID_BYTES)
        idbuf = (ctypes.c_uint8 * ID_BYTES).from_buffer_copy(unique_id)
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().smesh_comm_create(self.device, self.world, self.rank, idbuf, ctypes.byref(h)))
        self._h = h

    @staticmethod
    def unique_id():
        buf = (ctypes.c_uint8 * ID_BYTES)()
        _lib.check(_lib.lib().smesh_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def from_env(cls, device=None, port_offset=317):
        """One process per GPU, launched with RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT in the environment
        (what `python -m torch.distributed.run` and most launchers set).  The unique id is exchanged on
        MASTER_PORT + port_offset (override: SMESH_COMM_PORT)."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(os.environ.get("SMESH_COMM_PORT", int(os.environ.get("MASTER_PORT", "29500")) + port_offset))
        uid = cls.unique_id() if rank == 0 else None
        uid = exchange_id(uid, rank, world, addr, port)
        c = cls(device, rank, world, uid)
        got = c.nranks()
        if got != (rank, world):
            raise RuntimeError("RCCL communicator spans rank %d of %d, the launcher said rank %d of %d" % (got + (rank, world)))
        return c

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h is not None and h.value:
            try:
                _lib.lib().smesh_comm_destroy(h)
            except Exception:
                pass

    def allreduce(self, aggregator):
        """Sum the raw accumulators of all ranks in place in HBM (asynchronous, on the library stream)."""
        comms = (ctypes.c_void_p * 1)(self._h.value)
        aggs = (ctypes.c_void_p * 1)(aggregator._h.value)
        _lib.check(_lib.lib().smesh_allreduce(comms, aggs, 1))
        return aggregator

    def allreduce_rows(self, aggregator, row_lo, row_hi):
        """`allreduce` for the rows [row_lo, row_hi) only, on the library's exchange stream: it starts when what is queued on the main
        stream so far has finished and runs beside whatever the main stream is given next (`MeshAggregator.fuse_views_ranged` calls it
        through `on_rows`).  The main stream picks the result up at the aggregator's next use (`get()` ...)."""
        _lib.check(_lib.lib().smesh_allreduce_rows(self._h, aggregator._h, int(row_lo), int(row_hi)))
        return aggregator

    def reduce_scatter(self, aggregator):
        """Opt-in alternative to `allreduce` (`smesh_reduce_scatter`: one in-place ncclReduceScatter, half the bytes per link).
        Returns `(row_lo, row_hi)`: the rows of THIS rank's accumulator that now hold the sum over all ranks -- fetch them
        with `aggregator.get_rows(row_lo, row_hi)`; the other rows keep this rank's partial sums."""
        lo, hi = ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(_lib.lib().smesh_reduce_scatter(self._h, aggregator._h, ctypes.byref(lo), ctypes.byref(hi)))
        return int(lo.value), int(hi.value)

    def nranks(self):
        """(rank, number of ranks) as the RCCL communicator itself reports them."""
        rk, nr = ctypes.c_int(), ctypes.c_int()
        _lib.check(_lib.lib().smesh_comm_rank(self._h, ctypes.byref(rk), ctypes.byref(nr)))
        return int(rk.value), int(nr.value)

    def reduce_scalars(self, values, op="sum"):
        """Blocking reduction of a few host floats over all ranks (`op` sum | max)."""
        _lib.flush_pending(self.device)      # (the collective queues behind everything on the library stream: deferred views included)
        vals = [float(v) for v in values]
        buf = (ctypes.c_double * len(vals))(*vals)
        _lib.check(_lib.lib().smesh_comm_allreduce_f64(self._h, buf, len(vals), {"sum": 0, "max": 2}[op]))
        return list(buf)

    def barrier(self):
        """All ranks' library streams have drained and every rank has arrived."""
        self.reduce_scalars([1.0])


def create_all(devices):
    """Single process driving several GPUs: one communicator per device (`smesh_comm_create_all`)."""
    devs = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
    out = (ctypes.c_void_p * len(devices))()
    _lib.check(_lib.lib().smesh_comm_create_all(devs, len(devices), out))
    comms = []
    for i, d in enumerate(devices):
        c = Communicator.__new__(Communicator)
        c.device, c.rank, c.world, c._h = int(d), i, len(devices), ctypes.c_void_p(out[i])
        comms.append(c)
    return comms


def allreduce_all(comms, aggregators):
    """`smesh_allreduce` over all (communicator, aggregator) pairs this process holds, grouped into one RCCL call."""
    n = len(comms)
    c = (ctypes.c_void_p * n)(*[x._h.value for x in comms])
    a = (ctypes.c_void_p * n)(*[x._h.value for x in aggregators])
    _lib.check(_lib.lib().smesh_allreduce(c, a, n))
