"""Channel client for backends: the surface of /root/reference/frontend_connector.py on top of a pluggable
request/reply transport.

What a backend sees is unchanged -- `frontend_connector(parent_uuid, redis_channelizer_manager)`,
`create_channel(rate, freq) -> (channel_id, port)` with `port` a STRING as the reference returns it and
`(False, False)` on failure, `release_channel()`, `report_offset(x)`, `scan_mode_set_freq(f)`, `exit()`, the
attributes `.host`, `.channel_id`, `.channel_port`, `.my_client_id`, `.frequency` -- and so are the wire strings
(tests/golden/protocol.json was captured from the reference: `connect`, `create,<id>,<rate>,<freq>`,
`release,<id>,<chan>`, `offset,<id>,<chan>,<x>`, `hb,<id>`, `quit,<id>`, `scan_mode_set_freq,<f>`).

Inside it is organised differently from the reference: one `_Link` object owns the socket (pyzmq REQ with the
reference's 1 s timeouts when no factory is injected, else anything with send_string / recv_string, e.g.
rcf.protocol.LoopbackTransport), `_call(verb, *fields)` formats a request and splits the reply, and each public
method is a few lines on top of it.
"""
from __future__ import annotations

import logging
import threading
import time
import uuid

_TRIES = 5                       # frontend_connector.py:75-96: five attempts in each direction
_HEARTBEAT_S = 0.25              # frontend_connector.py:197-229


class _Link:
    """the REQ end of the channelizer protocol"""

    def __init__(self, host, port, factory):
        self.host = host
        self.ctx = None
        if factory is not None:
            self.sock = factory(host, port)
            return
        import zmq
        self.ctx = zmq.Context()
        self.sock = self.ctx.socket(zmq.REQ)
        for opt, val in ((zmq.RCVTIMEO, 1000), (zmq.SNDTIMEO, 1000), (zmq.LINGER, 0)):
            self.sock.setsockopt(opt, val)
        self.sock.connect("tcp://%s:%s" % (host, port))

    def close(self):
        for victim, how in ((self.sock, "close"), (self.ctx, "term"), (self.ctx, "destroy")):
            try:
                getattr(victim, how)()
            except Exception:
                pass


class frontend_connector():
    def __init__(self, parent_instance_uuid, redis_channelizer_manager, transport_factory=None,
                 heartbeat=True):
        self.log = logging.getLogger("%s.frontend_connector" % uuid.uuid4())
        self.redis_channelizer_manager = redis_channelizer_manager
        self.transport_factory = transport_factory
        self.thread_lock = threading.Lock()          # one conversation at a time
        self.send_lock = threading.Lock()            # one socket operation at a time
        self.continue_running = True
        self._link = None
        self.host = self.my_client_id = self.channel_id = self.frequency = self.last_create_channel = None
        self.channel_port = 0
        if heartbeat:
            threading.Thread(target=self.connection_handler, name="connection_handler", daemon=True).start()

    # the reference exposes these two; a few callers poke at them
    @property
    def socket(self):
        return self._link.sock if self._link else None

    @property
    def context(self):
        return (self._link.ctx or self._link) if self._link else None

    # ------------------------------------------------------------------ link management
    def connection_init(self, frequency):
        """pick the channelizer that covers `frequency` (nearest centre) and open a fresh link to it"""
        host, port = self.redis_channelizer_manager.get_channelizer_for_frequency(frequency)
        self._link = _Link(host, port, self.transport_factory)
        self.host = host
        self.my_client_id, self.channel_id, self.channel_port = None, None, 0

    def connection_teardown(self):
        if self._link is not None:
            self._link.close()

    def _attempt(self, what, op):
        """run one socket operation under the send lock; on failure log it the reference's way"""
        try:
            with self.send_lock:
                return True, op()
        except Exception as e:
            self.log.error("Exception in frontend_connector.%s(): %s %s" % (what, type(e), e))
            return False, None

    def send(self, data):
        """one request, one reply (split on commas), sharing a budget of five failures; None when it runs out"""
        failures = 0
        while failures < _TRIES:
            ok, _ = self._attempt("send", lambda: self.socket.send_string(data))
            if ok:
                break
            failures += 1
        while failures < _TRIES:
            ok, reply = self._attempt("recv", lambda: self.socket.recv_string())
            if ok:
                return reply.split(",")
            failures += 1
        return None

    def _call(self, verb, *fields):
        return self.send(",".join([verb] + [str(f) for f in fields]))

    @staticmethod
    def _is(reply, verb):
        return reply is not None and reply[0] == verb

    def connect(self):
        reply = self._call("connect")
        if reply is not None:
            self.my_client_id = int(reply[1])

    # ------------------------------------------------------------------ the calls backends make
    def scan_mode_set_freq(self, freq):
        with self.thread_lock:
            reply = self._call("scan_mode_set_freq", freq)
        # frontend_connector.py:122 compares the split LIST with the string 'success': never equal, so the
        # reference always answers False.  Kept bug-compatible (callers ignore the value).
        return reply == "success"

    def create_channel(self, channel_rate, freq):
        self.last_create_channel, self.frequency = time.time(), freq
        self.connection_init(freq)
        self.connect()
        with self.thread_lock:
            reply = self._call("create", self.my_client_id, channel_rate, freq)
            if self._is(reply, "create"):
                self.channel_id, self.channel_port = reply[1], reply[2]
                return self.channel_id, self.channel_port
            if reply is None or reply[0] == "na":
                self.log.error("Failed to create channel")
            return False, False

    def release_channel(self):
        with self.thread_lock:
            if self.channel_id is None:
                return False
            reply = self._call("release", self.my_client_id, self.channel_id)
            self.frequency = None
            if self._is(reply, "release"):
                self.channel_id = None
                return reply[1]
            if reply is None or reply[0] == "na":
                self.log.error("Failed to release channel, probably leaking channels")
            return False

    def report_offset(self, offset):
        with self.thread_lock:
            if self.channel_id is None:
                return False
            reply = self._call("offset", self.my_client_id, self.channel_id, offset)
            if self._is(reply, "offset"):
                return True
            if reply is None or reply[0] == "na":
                self.log.error("Failed to set offset")
                return False
            return None                               # any other verb: the reference falls off the end

    def exit(self):
        self.continue_running = False

    # ------------------------------------------------------------------ keep-alive
    def heartbeat_once(self):
        reply = self._call("hb", self.my_client_id)
        if reply is not None and reply[0] != "fail":
            return True
        self.log.error("Failed to heartbeat")
        self.connection_teardown()                    # the channelizer forgot us (or is gone): start over
        self.connection_init(self.frequency)
        self.connect()
        return False

    def connection_handler(self):
        time.sleep(0.1)
        while self.continue_running:
            if self._link is None or self.host is None:
                time.sleep(0.01)
                continue
            with self.thread_lock:
                try:
                    self.heartbeat_once()
                except Exception as e:
                    self.log.error("Failed to heartbeat: %s" % e)
            time.sleep(_HEARTBEAT_S)
        with self.thread_lock:
            try:
                self._call("quit", self.my_client_id)
                self.socket.close()
            except Exception:
                pass


def main(argv=None):
    """The reference's own smoke / speed test of this API (frontend_connector.py:232-251: it cannot run there any more --
    the constructor grew arguments), working: create + release one channel, then time 100 x create / release, against a
    running channelizer found through the registry.
        python -m rcf.frontend_connector [--registry redis | dir:<path>] [--transport zmq | tcp] [--freq 855000000] [--rate 25000]"""
    import argparse
    import sys
    from . import registry, transport
    ap = argparse.ArgumentParser(prog="python -m rcf.frontend_connector")
    ap.add_argument("--registry", default="redis")
    ap.add_argument("--transport", choices=["zmq", "tcp"], default="zmq")
    ap.add_argument("--freq", type=int, default=855000000)
    ap.add_argument("--rate", type=int, default=25000)
    ap.add_argument("-n", type=int, default=100)
    a = ap.parse_args(argv)
    mgr = registry.redis_channelizer_manager(clients=[transport.registry_client(a.registry)], start_thread=False)
    t0 = time.time()
    while not mgr.channelizers and time.time() - t0 < 10:
        mgr.poll_once()
        time.sleep(0.1)
    factory = transport.tcp_req_factory if a.transport == "tcp" else None
    test = frontend_connector("smoke-test", mgr, transport_factory=factory, heartbeat=False)
    channel_id, port = test.create_channel(a.rate, a.freq)
    if channel_id is False:
        raise Exception("test failed: create")
    if test.release_channel() is False:
        raise Exception("test failed: release")
    print("function test pass")
    start = time.time()
    for _ in range(a.n):
        test.create_channel(a.rate, a.freq)
        test.release_channel()
    print("speed test %s" % (time.time() - start))
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(main())
