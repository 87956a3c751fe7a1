"""CPU tests of the host side of the native communicator (semantic_meshes_amd/comm.py): the unique-id bootstrap over
plain sockets with world_size 3 (threads stand in for ranks; the RCCL calls themselves need GPUs and are covered by
tests/test_gpu_multi.py)."""
import socket
import threading

import pytest

from semantic_meshes_amd import comm


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_exchange_id_three_ranks():
    port = _free_port()
    payload = bytes(range(128))
    got = {}

    def run(rank):
        got[rank] = comm.exchange_id(payload if rank == 0 else None, rank, 3, "127.0.0.1", port, timeout=30.0)

    threads = [threading.Thread(target=run, args=(r,)) for r in (2, 1, 0)]   # peers first: they retry until rank 0 listens
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert got == {0: payload, 1: payload, 2: payload}


def test_exchange_id_skips_a_busy_port_and_foreign_listeners():
    port = _free_port()
    squatter = socket.socket()
    squatter.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    squatter.bind(("127.0.0.1", port))
    squatter.listen(4)                      # accepts, never answers with the job's handshake

    def deaf():
        try:
            while True:
                c, _ = squatter.accept()
                c.close()
        except OSError:
            pass

    threading.Thread(target=deaf, daemon=True).start()
    payload = b"x" * 128
    got = {}

    def run(rank):
        got[rank] = comm.exchange_id(payload if rank == 0 else None, rank, 2, "127.0.0.1", port, timeout=30.0)

    threads = [threading.Thread(target=run, args=(r,)) for r in (1, 0)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    squatter.close()
    assert got == {0: payload, 1: payload}


def test_exchange_id_world_one_and_timeout():
    assert comm.exchange_id(b"abc", 0, 1) == b"abc"
    with pytest.raises(TimeoutError):
        comm.exchange_id(None, 1, 2, "127.0.0.1", _free_port(), timeout=0.5)
