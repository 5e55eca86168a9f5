"""Narrowband egress: what /root/reference/rc_frontend/channel.py:36 does with
`zeromq.pub_sink(gr.sizeof_gr_complex, 1, 'tcp://0.0.0.0:<port>')` -- bare cf32 bytes on a ZMQ PUB socket,
no framing, lossy at the HWM -- fed from the channel rings of librcf instead of a GNU Radio flowgraph.

One pump thread per receiver: every `period` seconds it drains each in-use channel's ring
(`rcf_chan_read_iq`) and publishes the bytes on that channel's socket.  Backends keep using
`zeromq.sub_source('tcp://<connector.host>:<port>')` unchanged (p25_control_demod.py:244,
logging_receiver.py:98-99).  The socket factory is injectable; with pyzmq installed the default binds
real PUB sockets.  A second, optional socket per channel carries `quadrature_demod_cf(gain)` output
for consumers that want the discriminator done on the GPU.
"""
from __future__ import annotations

import logging
import threading
import time

log = logging.getLogger("egress")


def zmq_pub_factory():
    import zmq
    ctx = zmq.Context.instance()

    def make(port):
        s = ctx.socket(zmq.PUB)
        s.bind("tcp://0.0.0.0:%s" % port)
        return s
    return make


class EgressPump:
    def __init__(self, tb, socket_factory=None, period=0.01, fm_gain=None, fm_port_offset=1):
        """tb: rcf.receiver.receiver.  socket_factory(port) -> object with send(bytes) / close()."""
        self.tb = tb
        self.make = socket_factory or zmq_pub_factory()
        self.period = period
        self.fm_gain = fm_gain
        self.fm_port_offset = fm_port_offset
        self._plans = {}
        self.batch_cap = 1 << 13                         # samples per channel and pass in the batched read (0.3 s at 25 kS/s)
        self.socks = {}
        self.fm_socks = {}
        self.continue_running = True
        self.bytes_out = 0
        self.errors = 0
        self._thread = None
        self.prebound = {}                               # port -> (iq socket, fm socket or None)
        # bind when the channel is created, as the reference does (the PUB sink is part of the channel flowgraph,
        # channel.py:36, so a port in use fails channel construction and receiver.py:322-329 picks another)
        tb.bind_port = self.bind_port
        tb.release_port = self.release_port

    def bind_port(self, port):
        """-> True when the channel's socket(s) could be bound on `port`"""
        try:
            iq = self.make(port)
        except Exception as e:
            log.error("Failed to bind channel port %s: %s" % (port, e))
            return False
        fm = None
        if self.fm_gain is not None:
            try:
                fm = self.make(port + self.fm_port_offset)
            except Exception as e:
                log.error("Failed to bind fm port %s: %s" % (port + self.fm_port_offset, e))
                iq.close()
                return False
        self.prebound[port] = (iq, fm)
        return True

    def release_port(self, port):
        """close the sockets bind_port(port) bound when no channel came of it (construction failed, or the channel
        was destroyed before a pump pass adopted them)"""
        for s in self.prebound.pop(port, (None, None)):
            if s is not None:
                try:
                    s.close()
                except Exception:
                    pass

    def pump_once(self):
        with self.tb.access_lock:
            chans = dict(self.tb.channels)
            live_ports = {ch.port for ch in chans.values() if ch.chan_id is not None}
            for port in [p for p in self.prebound if p not in live_ports]:   # bound, but the channel is gone
                self.release_port(port)
        ready = []                                       # (block_id, iq, fm)
        with self.tb.access_lock:                        # destroy() / the idle sweep run under the same lock
            groups = {}
            for block_id, ch in chans.items():
                # one channel's failure (destroyed between the snapshot and the read, port already bound, reader
                # error) must not end the pump for everybody else: log, drop that channel's sockets, go on
                try:
                    if ch.chan_id is None:
                        continue
                    if block_id not in self.socks:
                        iq_s, fm_s = self.prebound.pop(ch.port, (None, None))
                        self.socks[block_id] = iq_s if iq_s is not None else self.make(ch.port)
                        if self.fm_gain is not None:
                            self.fm_socks[block_id] = fm_s if fm_s is not None else self.make(ch.port + self.fm_port_offset)
                    fe = getattr(ch, "frontend", None)       # channel-like objects without one are read singly
                    groups.setdefault(id(fe) if fe is not None else ("solo", block_id), (fe, []))[1].append((block_id, ch))
                except Exception as e:
                    self._fail(block_id, e)
            for fe, items in groups.values():
                ready.extend(self.read_group(fe, items))
        for block_id, iq, fm in ready:
            try:
                if len(iq):
                    payload = iq.tobytes()               # raw gr_complex items, arbitrary chunking
                    self.socks[block_id].send(payload)
                    self.bytes_out += len(payload)
                if fm is not None and len(fm) and block_id in self.fm_socks:
                    self.fm_socks[block_id].send(fm.tobytes())
            except Exception as e:
                self._fail(block_id, e)
        for block_id in [b for b in self.socks if b not in chans]:   # destroyed channels
            self._drop(block_id)

    def read_group(self, fe, items):
        """items: [(block_id, channel)] of one front-end -> [(block_id, iq, fm or None)] (called under tb.access_lock)"""
        ready = []
        iqs = fms = None
        if len(items) > 1 and hasattr(fe, "chan_read_many"):
            # all channels of one front-end behind ONE device synchronisation (rcf_chan_read_many).  The two
            # streams have independent reader positions: a batch that succeeded has ADVANCED its positions and
            # its samples are kept whatever happens to the other one -- only the stream whose batch failed is
            # read again channel by channel
            ids = tuple(ch.chan_id for _, ch in items)
            try:
                iqs = self._read_many(fe, ids, "iq", 1.0)
            except Exception as e:
                log.error("batched IQ egress read failed (%s): reading channel by channel" % e)
            if self.fm_gain is not None:
                try:
                    fms = self._read_many(fe, ids, "fm", self.fm_gain)
                except Exception as e:
                    log.error("batched discriminator egress read failed (%s): reading channel by channel" % e)
        for i, (block_id, ch) in enumerate(items):
            try:
                iq = iqs[i] if iqs is not None and iqs[i] is not None else ch.read_iq()
                if fms is not None and fms[i] is not None:
                    fm = fms[i]
                else:
                    fm = ch.read_fm(self.fm_gain) if block_id in self.fm_socks else None
                ready.append((block_id, iq, fm))
            except Exception as e:
                self._fail(block_id, e)
        return ready

    def _read_many(self, fe, ids, what, gain):
        """all channels of one front-end behind one device synchronisation; the call's arguments are kept while the
        channel set stays the same (native.Frontend.chan_read_many_plan)"""
        if not hasattr(fe, "chan_read_many_plan"):
            return fe.chan_read_many(list(ids), what, gain=gain, cap_each=self.batch_cap)
        key = (id(fe), what)
        plan = self._plans.get(key)
        if plan is None or plan[0] != (ids, gain):
            plan = ((ids, gain), fe.chan_read_many_plan(list(ids), what, gain=gain, cap_each=self.batch_cap))
            self._plans[key] = plan
        counts, out = plan[1]()
        return [None if counts[i] < 0 else out[i, :counts[i]].copy() for i in range(len(ids))]

    def _fail(self, block_id, e):
        self.errors += 1
        log.error("egress of channel %s failed: %s" % (block_id, e))
        self._drop(block_id)

    def _drop(self, block_id):
        for table in (self.socks, self.fm_socks):
            s = table.pop(block_id, None)
            if s is not None:
                try:
                    s.close()
                except Exception:
                    pass

    def run(self):
        while self.continue_running:
            try:
                self.pump_once()
            except Exception as e:                       # never let the daemon thread die silently
                self.errors += 1
                log.error("egress pump: %s" % e)
            time.sleep(self.period)

    def start(self):
        self._thread = threading.Thread(target=self.run, name="egress_pump")
        self._thread.daemon = True
        self._thread.start()
        return self

    def stop(self):
        self.continue_running = False
        if self._thread is not None:
            self._thread.join(timeout=2)
        # the channels' sockets (and the ones bound ahead of a channel) go with the pump: a PUB socket left bound keeps
        # its port for the life of the process (found by the ZeroMQ stand-in of tests/test_zmq_redis_branches.py)
        if self._thread is None or not self._thread.is_alive():
            for block_id in list(self.socks):
                self._drop(block_id)
            for port in list(self.prebound):
                self.release_port(port)
