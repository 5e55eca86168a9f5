"""Plain-socket stand-ins for the three services around the channelizer, for hosts without pyzmq / redis-py (this
build image has neither) and for tests that run daemon and client as two real processes:

  control wire   TcpRepServer / tcp_req_factory   the REQ/REP conversation of /root/reference/rc_frontend/receiver.py:
                 686-699 and frontend_connector.py:41-60: one UTF-8 request, one UTF-8 reply, strictly alternating per
                 connection; frames are a 4-byte little-endian length + the string (ZeroMQ's own framing is not spoken
                 -- the CSV strings inside are the reference's, byte for byte)
  data wire      TcpPubSocket / TcpSubSocket      what zeromq.pub_sink(itemsize 8, 'tcp://0.0.0.0:<port>') carries
                 (rc_frontend/channel.py:36): bare cf32 item bytes, no framing, arbitrary chunking, lossy for a
                 subscriber that does not keep up (a whole send is dropped for it, never part of one: the stream stays
                 item-aligned), no back-pressure on the publisher
  registry       DirRegistryClient                the six Redis commands redis_channel_publisher.py:63-90 and
                 redis_channelizer_manager.py:78-124 use (sadd / set / smembers / get / srem / delete) on a directory:
                 one file per key, atomic replace

With pyzmq / redis-py installed the real services are used instead (protocol.FrontendServer.serve_zmq,
egress.zmq_pub_factory, registry's default clients); `python -m rcf.frontend --transport tcp --registry dir:<path>`
selects these.  None of this touches the data path's arithmetic.
"""
from __future__ import annotations

import errno
import json
import os
import select
import selectors
import socket
import struct
import time


# ----------------------------------------------------------------------------------------------- control wire
def _send_frame(sock, s: str):
    data = s.encode("utf-8")
    sock.sendall(struct.pack("<I", len(data)) + data)


class _FrameReader:
    """incremental reader of length-prefixed frames from a non-blocking socket"""

    def __init__(self):
        self.buf = bytearray()

    def feed(self, chunk):
        self.buf += chunk

    def pop(self):
        if len(self.buf) < 4:
            return None
        (n,) = struct.unpack_from("<I", self.buf, 0)
        if n > (1 << 20):
            raise ValueError("control frame of %d bytes" % n)
        if len(self.buf) < 4 + n:
            return None
        s = bytes(self.buf[4:4 + n]).decode("utf-8", "replace")
        del self.buf[:4 + n]
        return s


class TcpRepServer:
    """The REP end: accepts any number of REQ connections, answers each request with handler(msg).  `poll(timeout)`
    serves whatever is ready and returns; `serve(server, stop)` is the reference's main loop around it (tick + recv +
    handle + send, receiver.py:620-699)."""

    OUT_LIMIT = 1 << 20                                    # a client that does not read its replies is dropped at this backlog

    def __init__(self, host="0.0.0.0", port=0):
        self.lsock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.lsock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        self.lsock.bind((host, port))
        self.lsock.listen(256)
        self.lsock.setblocking(False)
        self.port = self.lsock.getsockname()[1]
        self.endpoint = "tcp://%s:%d" % (host, self.port)
        self.conns = {}                                    # socket -> _FrameReader
        self.out = {}                                      # socket -> bytearray of reply bytes not yet written
        # epoll / kqueue where there is one: no limit of 1024 descriptors (select() raises above it), and the write set
        # costs nothing while nobody has a backlog
        self.sel = selectors.DefaultSelector()
        self.sel.register(self.lsock, selectors.EVENT_READ)

    def _flush(self, s):
        """write what the socket takes of its pending replies, NEVER blocking: one client that stops reading must not
        stall the control loop (heartbeat expiry, idle sweep, every other client) behind a send timeout"""
        buf = self.out.get(s)
        if not buf:
            return
        try:
            n = s.send(buf)
        except (BlockingIOError, InterruptedError):
            n = 0
        except OSError:
            self._drop(s)
            return
        del buf[:n]
        if len(buf) > self.OUT_LIMIT:
            self._drop(s)
            return
        want = selectors.EVENT_READ | (selectors.EVENT_WRITE if buf else 0)
        try:
            self.sel.modify(s, want)
        except (KeyError, ValueError, OSError):
            self._drop(s)

    def poll(self, handler, timeout=0.001):
        try:
            events = self.sel.select(timeout)
        except (OSError, ValueError):
            events = []
        served = 0
        for key, mask in events:
            s = key.fileobj
            if s is self.lsock:
                try:
                    c, _ = self.lsock.accept()
                except OSError:
                    continue
                c.setblocking(False)
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.conns[c] = _FrameReader()
                self.out[c] = bytearray()
                try:
                    self.sel.register(c, selectors.EVENT_READ)
                except (ValueError, OSError):
                    self._drop(c)
                continue
            if s not in self.conns:
                continue
            if mask & selectors.EVENT_WRITE:
                self._flush(s)
                if s not in self.conns:
                    continue
            if not (mask & selectors.EVENT_READ):
                continue
            try:
                chunk = s.recv(65536)
            except (BlockingIOError, InterruptedError):
                continue
            except OSError:
                chunk = b""
            if not chunk:
                self._drop(s)
                continue
            rd = self.conns[s]
            rd.feed(chunk)
            try:
                while True:
                    msg = rd.pop()
                    if msg is None:
                        break
                    reply = handler(msg)
                    data = (reply if reply is not None else "").encode("utf-8")
                    self.out[s] += struct.pack("<I", len(data)) + data
                    served += 1
            except Exception:
                self._drop(s)
                continue
            self._flush(s)
        return served

    def _drop(self, s):
        self.conns.pop(s, None)
        self.out.pop(s, None)
        try:
            self.sel.unregister(s)
        except (KeyError, ValueError, OSError):
            pass
        try:
            s.close()
        except OSError:
            pass

    def close(self):
        for s in list(self.conns):
            self._drop(s)
        try:
            self.sel.unregister(self.lsock)
        except (KeyError, ValueError, OSError):
            pass
        self.sel.close()
        self.lsock.close()


def serve_tcp(server, rep: TcpRepServer, stop=lambda: False, log=None):
    """rcf.protocol.FrontendServer behind a TcpRepServer: the body of the reference's `while 1:` (receiver.py:620-699)"""
    def handle(msg):
        try:
            return server.handle(msg)
        except Exception as e:                               # the reference logs and keeps serving
            if log is not None:
                log.error("handler error on %r: %s" % (msg, e))
            return "na"
    while not stop():
        server.tick()
        rep.poll(handle, 0.001)


class TcpReqSocket:
    """The REQ end with the reference client's socket options: 1 s send / receive timeouts, no linger
    (frontend_connector.py:45-52).  send_string / recv_string / close: what rcf.frontend_connector._Link asks of a
    transport."""

    def __init__(self, host, port, timeout=1.0):
        self.sock = socket.create_connection((host, int(port)), timeout=timeout)
        self.sock.settimeout(timeout)
        self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self.rd = _FrameReader()

    def send_string(self, s):
        _send_frame(self.sock, s)

    def recv_string(self):
        while True:
            msg = self.rd.pop()
            if msg is not None:
                return msg
            chunk = self.sock.recv(65536)                    # socket.timeout after 1 s, like zmq.Again on RCVTIMEO
            if not chunk:
                raise ConnectionError("channelizer closed the control connection")
            self.rd.feed(chunk)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


def tcp_req_factory(host, port):
    return TcpReqSocket(host, port)


# -------------------------------------------------------------------------------------------------- data wire
class TcpPubSocket:
    """PUB end of one channel: listens on `port`, every connected subscriber gets every send() -- or none of it, if its
    socket buffer cannot take the whole payload right now (ZeroMQ drops at the high-water mark too; the reference sets
    no back-pressure anywhere: channel.py:36).  Nothing is sent before a subscriber connects, as with PUB."""

    def __init__(self, port, host="0.0.0.0", sndbuf=1 << 20):
        self.lsock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        self.lsock.bind((host, int(port)))                   # no SO_REUSEADDR: a port in use must fail, receiver.py:322-329 retries
        self.lsock.listen(16)
        self.lsock.setblocking(False)
        self.port = self.lsock.getsockname()[1]
        self.subs = []
        self.sndbuf = sndbuf
        self.dropped = 0

    def _accept(self):
        while True:
            try:
                c, _ = self.lsock.accept()
            except (BlockingIOError, OSError):
                return
            c.setblocking(False)
            try:
                c.setsockopt(socket.SOL_SOCKET, socket.SO_SNDBUF, self.sndbuf)
            except OSError:
                pass
            self.subs.append([c, b""])

    def send(self, payload: bytes):
        self._accept()
        for sub in list(self.subs):
            c, pending = sub
            try:
                if pending:                                  # finish the message that was cut last time first
                    n = c.send(pending)
                    sub[1] = pending = pending[n:]
                    if pending:
                        self.dropped += 1                    # still stuck: this payload is dropped whole for this subscriber
                        continue
                n = c.send(payload)
                if n < len(payload):
                    sub[1] = payload[n:]                     # the tail goes out before anything newer: items stay aligned
            except BlockingIOError:
                self.dropped += 1
            except OSError:
                self.subs.remove(sub)
                try:
                    c.close()
                except OSError:
                    pass

    def close(self):
        for c, _ in self.subs:
            try:
                c.close()
            except OSError:
                pass
        self.subs = []
        self.lsock.close()


def tcp_pub_factory():
    return lambda port: TcpPubSocket(port)


class TcpSubSocket:
    """SUB end: what zeromq.sub_source('tcp://<host>:<port>') is to a backend (p25_control_demod.py:244): a stream of
    cf32 item bytes."""

    def __init__(self, host, port, timeout=5.0):
        deadline = time.time() + timeout
        while True:
            try:
                self.sock = socket.create_connection((host, int(port)), timeout=1.0)
                break
            except OSError:
                if time.time() > deadline:
                    raise
                time.sleep(0.02)
        self.sock.settimeout(timeout)

    def recv_exact(self, n_bytes):
        buf = bytearray()
        while len(buf) < n_bytes:
            chunk = self.sock.recv(min(1 << 20, n_bytes - len(buf)))
            if not chunk:
                raise ConnectionError("publisher closed after %d of %d bytes" % (len(buf), n_bytes))
            buf += chunk
        return bytes(buf)

    def close(self):
        try:
            self.sock.close()
        except OSError:
            pass


# --------------------------------------------------------------------------------------------------- registry
class DirRegistryClient:
    """sadd / set / smembers / get / srem / delete on a directory (keys = file names).  Members of a set are files in
    `<root>/<set>.set/`; values are files `<root>/<key>.val`, replaced atomically."""

    def __init__(self, root):
        self.root = root
        os.makedirs(root, exist_ok=True)

    def _val(self, key):
        return os.path.join(self.root, "%s.val" % _name(key))

    def _set(self, name):
        d = os.path.join(self.root, "%s.set" % _name(name))
        os.makedirs(d, exist_ok=True)
        return d

    def sadd(self, name, member):
        open(os.path.join(self._set(name), _name(member)), "w").close()
        return 1

    def srem(self, name, member):
        try:
            os.unlink(os.path.join(self._set(name), _name(member)))
            return 1
        except OSError:
            return 0

    def smembers(self, name):
        return {m.encode("utf-8") for m in os.listdir(self._set(name))}      # redis-py returns bytes

    def set(self, key, value):
        tmp = self._val(key) + ".tmp.%d" % os.getpid()
        with open(tmp, "w") as f:
            f.write(value if isinstance(value, str) else value.decode("utf-8"))
        os.replace(tmp, self._val(key))
        return True

    def get(self, key):
        try:
            with open(self._val(key)) as f:
                return f.read().encode("utf-8")
        except OSError as e:
            if e.errno == errno.ENOENT:
                return None
            raise

    def delete(self, key):
        try:
            os.unlink(self._val(key))
            return 1
        except OSError:
            return 0


def _name(key):
    key = key.decode("utf-8") if isinstance(key, bytes) else str(key)
    if "/" in key or key in ("", ".", ".."):
        raise ValueError("registry key %r" % key)
    return key


def registry_client(spec):
    """'redis' (redis-py, 127.0.0.1:6379 db 0: redis_channel_publisher.py:27,31) | 'dir:<path>' | 'none'"""
    if spec in (None, "none"):
        return None
    if spec.startswith("dir:"):
        return DirRegistryClient(spec[4:])
    if spec == "redis":
        import redis
        return redis.StrictRedis(host="127.0.0.1", port=6379, db=0)
    raise ValueError("registry %r" % spec)


def dumps_registry(client):
    """debug helper: {uuid: record} of everything in 'channelizers'"""
    out = {}
    for m in client.smembers("channelizers"):
        v = client.get(m)
        if v is not None:
            out[m.decode("utf-8") if isinstance(m, bytes) else m] = json.loads(v)
    return out
