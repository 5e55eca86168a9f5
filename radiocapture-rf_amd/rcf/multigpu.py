"""Multi-GPU glue: one process per GPU, one SDR front-end / spectrum slice each.  The hot path shards by
independent front-ends -- the reference already runs one channelizer process per SDR (systemd
radiocapture-channelizer@<i>, /root/reference/rc_frontend/receiver.py:67-70) -- so there is NO data-path
collective.  The only exchange is the all-gather of detected-peak lists after a scan (BASELINE.json configs[4]):

  * on GPUs it is ONE ncclAllGather over xGMI through the C ABI (`rcf_allgather_peaks`, librccl loaded by
    librcf on first use); the 128-byte communicator id travels over `HostGroup`, a plain TCP rendezvous
    (rank 0 listens next to MASTER_PORT) -- no PyTorch anywhere on this path;
  * `allgather_peaks_host` is the same exchange over the HostGroup alone (no GPU-to-GPU link, CPU-only tests).
(The world-size-2 gloo test carries the same records over torch.distributed with helpers of its own: tests/test_dist_gloo.py.)
"""
from __future__ import annotations

import socket
import struct
import time

import numpy as np

PEAK_CAP = 1024
_MAGIC = b"RCFG"


def sources_for_rank(n_sources: int, world: int, rank: int):
    """Front-end g -> rank g % world (one per GPU when n_sources == world)."""
    return [s for s in range(n_sources) if s % world == rank]


def route_frequency(freq, centers, samp_rates):
    """Same rule as receiver.connect_channel_xlat (receiver.py:288-291): the source whose centre is
    nearest among those with abs(f - center) < samp_rate / 2; None when out of band."""
    best, dist_ = None, None
    for g, (c, r) in enumerate(zip(centers, samp_rates)):
        d = abs(freq - c)
        if d < r / 2 and (dist_ is None or d < dist_):
            best, dist_ = g, d
    return best


def pack_peaks(freqs_hz, cap=PEAK_CAP):
    """Fixed-capacity record: [count, f0, f1, ...] int64, -1 padded (latency-bound 8 KiB payload)."""
    rec = np.full(cap + 1, -1, dtype=np.int64)
    n = min(len(freqs_hz), cap)
    rec[0] = n
    rec[1:1 + n] = np.asarray(freqs_hz[:n], dtype=np.int64)
    return rec


def unpack_peaks(records):
    out = []
    for rec in records:
        n = int(rec[0])
        out.extend(int(v) for v in rec[1:1 + n])
    return sorted(out)


# --------------------------------------------------------------------------- host rendezvous
def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed during a %d-byte read" % n)
        buf += chunk
    return bytes(buf)


def _send_msg(sock, data: bytes):
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv_msg(sock) -> bytes:
    (n,) = struct.unpack("<I", _recv_exact(sock, 4))
    return _recv_exact(sock, n)


class HostGroup:
    """Host-side rendezvous for `world` processes of one node: star over TCP, rank 0 in the middle.
    all_gather / broadcast / barrier of small byte strings -- enough to hand out the RCCL id, to line the ranks
    up, and (fallback) to exchange the peak lists themselves.  Rank 0 listens on the first free port of
    [port, port + 16); the others probe that range until a listener answers with the group's magic."""

    def __init__(self, rank, world, addr="127.0.0.1", port=29600, timeout=120.0):
        self.rank, self.world = int(rank), int(world)
        self.peers = {}
        self.sock = None
        if self.world <= 1:
            return
        deadline = time.time() + timeout
        if self.rank == 0:
            srv = None
            for p in range(port, port + 16):
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, p))
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise RuntimeError("HostGroup: no free port in [%d, %d)" % (port, port + 16))
            srv.listen(self.world)
            srv.settimeout(1.0)
            while len(self.peers) < self.world - 1:
                if time.time() > deadline:
                    raise TimeoutError("HostGroup: %d of %d ranks arrived" % (len(self.peers) + 1, self.world))
                try:
                    c, _ = srv.accept()
                except socket.timeout:
                    continue
                c.settimeout(timeout)
                try:
                    hello = _recv_exact(c, 8)
                except Exception:
                    c.close()
                    continue
                if hello[:4] != _MAGIC:
                    c.close()
                    continue
                r = struct.unpack("<I", hello[4:])[0]
                c.sendall(_MAGIC)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.peers[r] = c
            srv.close()
        else:
            while self.sock is None:
                if time.time() > deadline:
                    raise TimeoutError("HostGroup: rank %d found no rank 0 near port %d" % (self.rank, port))
                for p in range(port, port + 16):
                    try:
                        s = socket.create_connection((addr, p), timeout=1.0)
                        s.settimeout(5.0)
                        s.sendall(_MAGIC + struct.pack("<I", self.rank))
                        if _recv_exact(s, 4) == _MAGIC:
                            s.settimeout(timeout)
                            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                            self.sock = s
                            break
                        s.close()
                    except Exception:
                        continue
                if self.sock is None:
                    time.sleep(0.2)

    def all_gather(self, data: bytes):
        """-> list of every rank's bytes, in rank order"""
        if self.world <= 1:
            return [data]
        if self.rank == 0:
            parts = [data] + [b""] * (self.world - 1)
            for r, c in self.peers.items():
                parts[r] = _recv_msg(c)
            blob = b"".join(struct.pack("<I", len(p)) + p for p in parts)
            for c in self.peers.values():
                _send_msg(c, blob)
            return parts
        _send_msg(self.sock, data)
        blob = _recv_msg(self.sock)
        parts, at = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<I", blob, at)
            parts.append(blob[at + 4: at + 4 + n])
            at += 4 + n
        return parts

    def broadcast(self, data=None) -> bytes:
        """rank 0's bytes on every rank"""
        return self.all_gather(data if self.rank == 0 and data is not None else b"")[0]

    def barrier(self):
        self.all_gather(b"")

    def max(self, value: float) -> float:
        return max(struct.unpack("<d", p)[0] for p in self.all_gather(struct.pack("<d", float(value))))

    def close(self):
        for c in self.peers.values():
            c.close()
        self.peers = {}
        if self.sock is not None:
            self.sock.close()
            self.sock = None


def init_comm(frontend, group: HostGroup):
    """Give `frontend` (rcf.native.Frontend) an RCCL communicator spanning the group: rank 0 draws the id,
    the HostGroup carries it, every rank joins (collective)."""
    from . import native
    uid = native.comm_unique_id() if group.rank == 0 else None
    uid = group.broadcast(uid)
    frontend.comm_init(group.rank, group.world, uid)


# --------------------------------------------------------------------------- the exchange, two transports
def allgather_peaks(frontend, freqs_hz, cap=PEAK_CAP):
    """Every rank ends with the global sorted list of detected peak frequencies: ncclAllGather over xGMI via
    the C ABI (rcf_allgather_peaks) on the front-end's device."""
    parts = frontend.allgather_peaks(np.asarray(freqs_hz[:cap], dtype=np.int64), cap)
    return sorted(int(v) for p in parts for v in p)


def allgather_peaks_host(group: HostGroup, freqs_hz, cap=PEAK_CAP):
    """The same exchange over the host rendezvous (no GPU-to-GPU path / CPU-only runs)."""
    recs = group.all_gather(pack_peaks(freqs_hz, cap).tobytes())
    return unpack_peaks([np.frombuffer(r, dtype=np.int64) for r in recs])
