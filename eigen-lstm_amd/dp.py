"""Data-parallel decomposition of the window (host-side logic shared by bench.py and the tests).

The reference is single-device.  The build shards the B concurrent streams (batch columns) across
ranks: every op of the window is column-wise in B except the weight-gradient sums
(OV/lstm_eigen_opt/lstm.cc:271-272,297-299), so

    rank r owns streams [r*B/R, (r+1)*B/R) with their cursors and h/c carry;
    parameters and Adagrad memory are replicated;
    one SUM all-reduce of the flat gradient block [dW|dU|db|dWhy|dby] per window;
    every rank then applies the identical Adagrad step;
    the window loss is sum_r (local surprisal sum / GLOBAL batch).
"""
import numpy as np


def shard(rank, world, global_batch):
    """(first stream, stream count) of `rank`."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not a multiple of {world} ranks")
    per = global_batch // world
    return rank * per, per


def cursors(length, S, rank, world, global_batch):
    """pos[b] = S + (b_global*(len-S))/B_global for this rank's streams (deterministic stand-in for
    rand()%(len-S)+S, OV/lstm_eigen_opt/lstm.cc:140-144)."""
    first, per = shard(rank, world, global_batch)
    return np.array([S + ((first + b) * (length - S)) // global_batch for b in range(per)], dtype=np.uint64)


def local_loss_to_global(local_loss, local_batch, global_batch):
    """A rank that divided its surprisal sum by its LOCAL batch rescales before the sum over ranks."""
    return local_loss * (local_batch / global_batch)


# ---- rendezvous for `bench.py --gpus N` ------------------------------------------------------------------
# The ranks only ever exchange a 128-byte RCCL id, barriers and a few doubles.  A star over plain TCP
# sockets (rank 0 listens on an ephemeral port and publishes it in a file named after MASTER_PORT, the
# ranks being on ONE node by contract) keeps the GPU processes free of any second HIP/RCCL copy that
# importing torch would map next to the library's own.
import os
import pickle
import socket
import struct
import tempfile
import time


class Rendezvous:
    def __init__(self, rank, world, tag, timeout=300.0):
        self.rank, self.world = rank, world
        self.peers = []
        path = os.path.join(tempfile.gettempdir(), f"lstm_hip_rdzv_{tag}")
        if world == 1:
            return
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(world)
            tmp = path + f".{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, path)
            srv.settimeout(timeout)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                r = struct.unpack("<i", self._recv_exact(c, 4))[0]
                conns[r] = c
            self.peers = [conns[r] for r in range(1, world)]
            srv.close()
            try:
                os.unlink(path)
            except OSError:
                pass
        else:
            t0 = time.time()
            while True:
                try:
                    port = int(open(path).read())
                    c = socket.create_connection(("127.0.0.1", port), timeout=5.0)
                    break
                except (OSError, ValueError):
                    if time.time() - t0 > timeout:
                        raise TimeoutError(f"rank {rank}: no rendezvous at {path}")
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(timeout)
            c.sendall(struct.pack("<i", rank))
            self.peers = [c]

    @staticmethod
    def _recv_exact(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous peer closed")
            buf += part
        return buf

    def _send(self, c, obj):
        data = pickle.dumps(obj)
        c.sendall(struct.pack("<q", len(data)) + data)

    def _recv(self, c):
        n = struct.unpack("<q", self._recv_exact(c, 8))[0]
        return pickle.loads(self._recv_exact(c, n))

    def allgather(self, obj):
        """list of every rank's `obj`, in rank order (also a barrier)."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [self._recv(c) for c in self.peers]
            for c in self.peers:
                self._send(c, out)
            return out
        self._send(self.peers[0], obj)
        return self._recv(self.peers[0])

    def barrier(self):
        self.allgather(None)

    def broadcast(self, obj):
        return self.allgather(obj if self.rank == 0 else None)[0]

    def close(self):
        for c in self.peers:
            try:
                c.close()
            except OSError:
                pass
