"""Control wire of the channelizer: the CSV REQ/REP protocol of
/root/reference/rc_frontend/receiver.py:503-614 (server) and /root/reference/frontend_connector.py
(client), transport-agnostic.

`FrontendServer.handle(msg) -> reply` is the reference's nested `handler(msg, tb)` with its `clients` /
`client_hb` tables, `tick(now)` is the body of its main loop (10 s status + idle-channel sweep,
5 s client heartbeat expiry, receiver.py:620-680).  ZeroMQ is optional: `serve_zmq()` runs the same
non-blocking REP loop when pyzmq is importable; tests drive `handle()` directly.
"""
from __future__ import annotations

import logging
import time

log = logging.getLogger("frontend")


class FrontendServer:
    def __init__(self, tb, clock=time.time):
        self.tb = tb                      # rcf.receiver.receiver (or anything with its surface)
        self.clients = {}
        self.client_hb = {}
        self.client_num = 0
        self.clock = clock
        self.start_time = clock()
        self.last_status = clock()
        self.status_extra = None          # callable -> dict appended to the 10 s status line (the daemon sets it)

    # ------------------------------------------------------------------ receiver.py:503-614
    def handle(self, msg):
        tb = self.tb
        data = msg.strip().split(",")
        if data[0] == "create":
            c = int(data[1])
            channel_rate = int(data[2])
            freq = int(data[3])
            try:
                block_id, port = tb.connect_channel(channel_rate, freq)
            except Exception as e:
                block_id = -1
                log.error("Exception: %s" % e)
            if block_id == -1 or block_id is False:
                log.error("failed to create channel %s" % freq)
                return "na,%s" % freq
            try:
                self.clients[c].append(block_id)
            except Exception as e:       # 'create' before 'connect' (receiver.py:527-533)
                log.error("Exception in channel creation %s" % e)
                tb.release_channel(block_id)
                return "na,%s" % freq
            return "create,%s,%s" % (block_id, port)
        elif data[0] == "release":
            try:
                c = int(data[1])
                block_id = data[2]
                result = tb.release_channel(block_id)
                if result == -1:
                    return "na,%s" % block_id
                try:
                    self.clients[c].remove(block_id)
                except ValueError:
                    pass
                return "release,%s" % block_id
            except Exception:
                return "na\n"
        elif data[0] == "scan_mode_set_freq":
            freq = int(data[1])
            tb.scan_mode_set_freq(freq)   # an exception propagates, as in the reference (:564-566)
            return "success"
        elif data[0] == "quit":
            c = int(data[1])
            try:
                for x in self.clients[c]:
                    tb.release_channel(x)
            except KeyError:
                pass
            finally:
                self.client_hb.pop(c, None)
                self.clients.pop(c, None)
            return "quit,%s" % c
        elif data[0] == "connect":
            c = self.client_num
            self.client_num += 1
            self.clients[c] = []
            self.client_hb[c] = self.clock()
            return "connect,%s" % c
        elif data[0] == "hb":
            try:
                c = int(data[1])
            except Exception:
                return "fail,0"
            if c not in self.client_hb:
                return "fail,%s" % c
            self.client_hb[c] = self.clock()
            return "hb,%s" % c
        elif data[0] == "offset":
            client_id = int(data[1])
            block_id = data[2]
            offset = float(data[3])
            tb.source_offset(block_id, offset)
            return "offset,%s" % client_id
        return None                       # unknown verbs get no reply object in the reference either

    # ------------------------------------------------------------------ receiver.py:620-680
    def tick(self, now=None):
        now = self.clock() if now is None else now
        tb = self.tb
        if now - self.last_status > 10:
            # receiver.py:622-625, the 10 s status line (+ what SURVEY 5 asks to add to it: Msps in, kernel time)
            log.info("Frontend Status: client: %s client_hb: %s channels: %s uptime: %s" % (
                len(self.clients), len(self.client_hb), len(tb.channels), int(now - self.start_time)))
            if self.status_extra is not None:
                try:
                    log.info("Frontend Status: %s" % self.status_extra())
                except Exception as e:
                    log.error("status metrics failed: %s" % e)
            self.last_status = now
            if now - tb.last_channel_cleanup > tb.channel_idle_timeout * 2:
                tb.sweep_idle_channels(now)
        deletions = []
        for client in list(self.client_hb):
            if now - self.client_hb[client] > 5:
                log.warning("Client heartbeat timeout %s" % client)
                for x in self.clients.get(client, []):
                    tb.release_channel(x)
                self.clients[client] = []
                deletions.append(client)
        for c in deletions:
            self.client_hb.pop(c, None)
            self.clients.pop(c, None)
        return deletions

    def serve_zmq(self, bind="tcp://0.0.0.0:0", stop=lambda: False):
        """The reference's REP loop (receiver.py:686-699).  Needs pyzmq."""
        import zmq
        ctx = zmq.Context()
        sock = ctx.socket(zmq.REP)
        sock.bind(bind)
        self.endpoint = sock.getsockopt(zmq.LAST_ENDPOINT).decode("utf-8")
        try:
            while not stop():
                self.tick()
                try:
                    msg = sock.recv_string(flags=zmq.NOBLOCK)
                except zmq.Again:
                    time.sleep(0.001)
                    continue
                # the reference's main loop catches and logs whatever its handler raises and keeps serving
                # (receiver.py:686-699); a REP socket must also answer every request or it wedges
                try:
                    resp = self.handle(msg)
                except Exception as e:                   # malformed request, failed retune, ...
                    log.error("handler error on %r: %s" % (msg, e))
                    resp = "na"
                sock.send_string(resp if resp is not None else "")
        finally:
            # (the reference's loop never ends; this one does -- stop() -- and gives its port back)
            try:
                sock.close(0)
            finally:
                ctx.term()


class LoopbackTransport:
    """In-process REQ socket for tests and single-process deployments: send_string / recv_string."""

    def __init__(self, server):
        self.server = server
        self._pending = None

    def send_string(self, s):
        self._pending = self.server.handle(s)

    def recv_string(self):
        r, self._pending = self._pending, None
        if r is None:
            raise RuntimeError("no reply")
        return r

    def close(self):
        pass
