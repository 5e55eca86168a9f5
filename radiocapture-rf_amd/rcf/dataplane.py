"""The channelizer process's data plane on the native pump.

In the reference, receiver.start() (rc_frontend/receiver.py:271) hands ALL configured sources to GNU Radio's scheduler
threads (one top block, receiver.py:67-70,170-204; ten RTL-SDRs per host in configs/config_denver_dev_den817.py:25-118) and
every channel's pub_sink (channel.py:36) is fed by them: no Python touches a sample.  Here the same holds:

    source rings (pinned host memory, whole blocks)      one per configured source
        |   rcf_pump_t: ONE native thread per class of sources (same rate and wire format), no interpreter in it --
        |   whichever blocks are complete go out as one group block (one conversion / filterbank / FIR-bank / discriminator
        v   launch for all of them), the subscribed channels' new output is gathered by one more launch ...
    per-channel host rings (pinned; the gather kernel writes them across PCIe)
        |   ... and the egress thread below only copies ready spans out of them and calls socket.send()
        v
    PUB sockets (bare cf32 bytes, channel.py:36)

Sources:
  'synthetic'   the seeded tile in wire format IS the ring; the pump replays it at samp_rate of wall-clock time
  'file'        a feeder thread (rcf.sources.PacedSource) copies the capture into the ring block by block at samp_rate and
                counts the blocks complete (rcf_pump_config_t.written)
  anything else receiver.feed / feed_raw -- whatever owns the SDR hands samples over as they arrive -- lands in the ring
                through RingWriter the same way

Channels come and go under the running pump (rcf_pump_subscribe / rcf_pump_unsubscribe), as connect_channel /
release_channel + the idle sweep create and destroy channel flowgraphs under the reference's running top block.
"""
from __future__ import annotations

import logging
import math
import threading

import numpy as np

from . import egress, sources

log = logging.getLogger("dataplane")


def block_for(tile_samples, fs, block_ms):
    """the block a clock-paced ring is cut into: the longest whole fraction of the tile that is not longer than block_ms
    (the ring is replayed block by block, so the tile must be whole blocks)"""
    want = max(1, int(round(fs * block_ms * 1e-3)))
    m = max(1, int(math.ceil(tile_samples / want)))
    while tile_samples % m:
        m += 1
    return tile_samples // m


class RingWriter:
    """producer side of a counter-fed source ring: write(samples) copies into the ring (wrapping) and counts every block
    that became complete -- the pump takes block k from slot k % ring_blocks once the counter is > k.  Producers run at
    the sources' clock (an SDR, a paced file feeder): one that got more than ring_blocks - 1 blocks ahead of the pump would
    overwrite blocks not yet taken, as an SDR driver's own ring overruns when its reader stalls."""

    # (how many blocks a counter-fed ring holds is NativeDataPlane's ring_blocks, 64 = 1.3 s of 20 ms blocks: the first
    # blocks of a process take hundreds of milliseconds -- code objects load, buffers are touched for the first time -- and
    # the producer does not wait)
    def __init__(self, ring, block_items, ring_blocks):
        self.ring, self.block_items, self.ring_blocks = ring, int(block_items), int(ring_blocks)
        self.counter = np.zeros(1, dtype=np.uint64)
        self.at = 0                                    # items written so far
        self.lock = threading.Lock()

    def write(self, items):
        with self.lock:
            n, cap = len(items), self.block_items * self.ring_blocks
            done = 0
            while done < n:
                pos = self.at % cap
                take = min(n - done, cap - pos, self.block_items - (self.at % self.block_items))
                self.ring[pos:pos + take] = items[done:done + take]
                self.at += take
                done += take
                if self.at % self.block_items == 0:
                    self.counter[0] = self.at // self.block_items      # (the block is whole before it is counted)


class _Class:
    """the sources of one receiver that can share a pump: same rate, wire format and block"""

    def __init__(self, key):
        self.key = key
        self.fs, self.wire, self.block = key
        self.members = []              # (source id of tb.sources, front-end, config dict)
        self.rings, self.written, self.writers, self.feeders = [], [], {}, []
        self.group = self.pump = None


class NativeDataPlane(egress.EgressPump):
    """EgressPump whose channels are delivered by native pumps: the sockets, the port hooks and the per-channel failure
    handling are the parent's; what differs is who moves the samples (nobody in Python) and where a pass reads them
    (the pump's host rings instead of rcf_chan_read_many + a device synchronisation)."""

    def __init__(self, tb, socket_factory=None, period=0.01, fm_gain=None, block_ms=20.0, max_channels=1024,
                 out_ring_samples=1 << 13, ring_blocks=64, cpu=-1, spin_us=0, batch_window_s=0.0):
        super().__init__(tb, socket_factory=socket_factory, period=period, fm_gain=fm_gain)
        from . import native
        self.native = native
        self.block_ms, self.max_channels, self.out_ring_samples = float(block_ms), int(max_channels), int(out_ring_samples)
        self.ring_blocks, self.cpu, self.spin_us, self.batch_window_s = int(ring_blocks), int(cpu), int(spin_us), float(batch_window_s)
        self.classes = {}
        self.member_of = {}            # id(front-end) -> (class, member index)
        self.subs = {}                 # block_id -> [class, chan_id, iq slot, fm slot or None]
        self._plans = {}
        self._pins = []
        self._build()

    # ------------------------------------------------------------------ construction
    def _build(self):
        tb, native = self.tb, self.native
        if any("parent_chan" in s for s in tb.sources.values()):
            raise ValueError("receiver_split2 sources are fed in pieces (receiver.feed): no native pump for them")
        seen = set()
        for sid in sorted(tb.sources):
            src = tb.sources[sid]
            fe, real = src["block"], src["source_id"]
            if real in seen or not hasattr(fe, "_h"):
                if not hasattr(fe, "_h"):
                    raise ValueError("source %s has no native front-end" % sid)
                continue
            seen.add(real)
            cfg = tb.realsources[real]
            fs = float(cfg["samp_rate"])
            kind = cfg.get("type")
            wire = cfg.get("wire" if kind == "synthetic" else "format", "cf32") if kind in ("synthetic", "file") else cfg.get("wire", "cf32")
            if wire not in sources.WIRE:
                raise ValueError("source %s: wire format %r" % (sid, wire))
            bms = float(cfg.get("block_ms", self.block_ms))
            if kind == "synthetic":
                block = block_for(int(cfg.get("tile_samples", 1 << 20)), fs, bms)
            else:
                block = max(1, int(round(fs * bms * 1e-3)))
            cl = self.classes.setdefault((fs, wire, block), _Class((fs, wire, block)))
            self.member_of[id(fe)] = (cl, len(cl.members))
            cl.members.append((sid, fe, cfg))
        for cl in self.classes.values():
            dt, _ = sources.WIRE[cl.wire]
            per = 1 if cl.wire == "cf32" else 2
            for m, (sid, fe, cfg) in enumerate(cl.members):
                kind = cfg.get("type")
                if kind == "synthetic":
                    data = sources.to_wire(sources.synthetic_tile(cfg), cl.wire)
                    pin = native.PinnedArray(len(data), dt)
                    pin.array[:] = data
                    cl.written.append(None)
                else:
                    pin = native.PinnedArray(cl.block * per * self.ring_blocks, dt)
                    pin.array[:] = 0
                    w = RingWriter(pin.array, cl.block * per, self.ring_blocks)
                    cl.writers[sid] = w
                    cl.written.append(w.counter)
                    if kind == "file":
                        cl.feeders.append(_FileFeeder(sid, cfg, w, cl.block, per))
                self._pins.append(pin)
                cl.rings.append(pin)
        # receiver.feed / feed_raw of a pumped source go to its ring; a clock-paced one has no producer but the clock
        self._orig_feed, self._orig_feed_raw = tb.feed, tb.feed_raw
        tb.feed, tb.feed_raw = self._feed, self._feed_raw

    def _writer(self, source_id):
        for cl in self.classes.values():
            if source_id in cl.writers:
                return cl, cl.writers[source_id]
        raise RuntimeError("source %s is paced by the data plane's clock (type 'synthetic'): it takes no samples" % source_id)

    def _feed(self, source_id, iq):
        cl, w = self._writer(source_id)
        if cl.wire != "cf32":
            raise ValueError("source %s is configured for wire format %r: use feed_raw" % (source_id, cl.wire))
        w.write(np.ascontiguousarray(iq, dtype=np.complex64))
        self.tb._fed[source_id] = self.tb._fed.get(source_id, 0) + len(iq)

    def _feed_raw(self, source_id, raw, fmt, scale, offset=0.0):
        cl, w = self._writer(source_id)
        if sources.WIRE[cl.wire][1] != fmt:
            raise ValueError("source %s is configured for wire format %r" % (source_id, cl.wire))
        w.write(np.ascontiguousarray(raw, dtype=sources.WIRE[cl.wire][0]))
        self.tb._fed[source_id] = self.tb._fed.get(source_id, 0) + len(raw) // 2

    # ------------------------------------------------------------------ run
    def start(self):
        native = self.native
        for cl in self.classes.values():
            fes = [fe for _, fe, _ in cl.members]
            cl.group = native.Group(fes)
            scale, offset = sources.WIRE_SCALE.get(cl.wire, (1.0, 0.0))
            fmt = sources.WIRE[cl.wire][1] or native.FMT_CF32
            n = len(fes)
            period = cl.block / cl.fs
            cl.pump = native.Pump(cl.group, cl.rings, cl.block, cl.fs, (), fmt=fmt, scale=scale, offset=offset, what="iq",
                                  phase_s=[period * i / n for i in range(n)],      # the sources' blocks do not all end at once
                                  # (the first second is the process coming up -- code objects loading, first-touch
                                  # allocations: its blocks run but the late / latency statistics do not judge them)
                                  warm_blocks=int(math.ceil(1.0 / period)),
                                  out_ring_samples=self.out_ring_samples, max_read=self.max_channels, cpu=self.cpu,
                                  spin_us=self.spin_us, batch_window_s=self.batch_window_s, written=cl.written)
            for f in cl.feeders:
                f.start()
        return super().start()

    def detail(self):
        """every pump's own statistics (rcf_pump_stats_t), keyed by its class of sources"""
        return {"%g sps %s x %d" % cl.key: cl.pump.stats() for cl in self.classes.values() if cl.pump is not None}

    def stop(self):
        super().stop()
        try:
            for k, st in self.detail().items():
                log.info("pump [%s]: %s" % (k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in st.items()}))
        except Exception:
            pass
        for cl in self.classes.values():
            for f in cl.feeders:
                f.stop()
            if cl.pump is not None:
                cl.pump.stop()
                cl.pump = None
            if cl.group is not None:
                cl.group.close()
                cl.group = None
        self.tb.feed, self.tb.feed_raw = self._orig_feed, self._orig_feed_raw
        for p in self._pins:
            p.free()
        self._pins = []

    # ------------------------------------------------------------------ one egress pass over one front-end's channels
    def read_group(self, fe, items):
        where = self.member_of.get(id(fe)) if fe is not None else None
        if where is None or where[0].pump is None:
            return super().read_group(fe, items)
        cl, member = where
        for block_id, ch in items:
            sub = self.subs.get(block_id)
            if sub is not None and sub[1] != ch.chan_id:          # retuned: the object has a new native channel
                self._unsubscribe(block_id)
                sub = None
            if sub is None:
                try:
                    iq_slot = cl.pump.subscribe(member, ch.chan_id, "iq")
                    fm_slot = None
                    if self.fm_gain is not None:
                        try:
                            fm_slot = cl.pump.subscribe(member, ch.chan_id, "fm", self.fm_gain)
                        except Exception:
                            cl.pump.unsubscribe(iq_slot)
                            raise
                    self.subs[block_id] = [cl, ch.chan_id, iq_slot, fm_slot]
                except Exception as e:
                    self._fail(block_id, e)
        live = [(b, self.subs[b]) for b, _ in items if b in self.subs]
        if not live:
            return []
        slots = tuple(s[2] for _, s in live) + tuple(s[3] for _, s in live if s[3] is not None)
        key = id(fe)                                              # (one buffer per front-end: a pass sends after it has read them all)
        plan = self._plans.get(key)
        if plan is None or plan[0] != slots:
            plan = (slots, cl.pump.read_many_plan(list(slots), cap_each=self.out_ring_samples))
            self._plans[key] = plan
        got = plan[1]()
        ready, j = [], len(live)
        for i, (block_id, s) in enumerate(live):
            fm = None
            if s[3] is not None:
                fm = got[j]
                j += 1
            ready.append((block_id, got[i], fm))
        return ready

    def _unsubscribe(self, block_id):
        sub = self.subs.pop(block_id, None)
        if sub is None or sub[0].pump is None:
            return
        for slot in sub[2:]:
            if slot is not None:
                try:
                    sub[0].pump.unsubscribe(slot)
                except Exception as e:
                    log.error("unsubscribe of channel %s failed: %s" % (block_id, e))

    def _drop(self, block_id):
        self._unsubscribe(block_id)
        super()._drop(block_id)

    # ------------------------------------------------------------------ what the status line carries
    def stats(self):
        out = {"rcf_pump_classes": len(self.classes)}
        tot = {"blocks_done": 0, "late": 0, "overruns": 0, "group_blocks": 0, "samples_out": 0}
        worst_p99 = worst_max = 0.0
        err = None
        samples = 0
        for cl in self.classes.values():
            if cl.pump is None:
                continue
            st = cl.pump.stats()
            for k in tot:
                tot[k] += st[k]
            samples += st["blocks_done"] * cl.block
            self.tb._fed[("pump",) + cl.key] = st["blocks_done"] * cl.block      # (receiver.metrics: samples in, Msps)
            worst_p99, worst_max = max(worst_p99, st["latency_ms_p99"]), max(worst_max, st["latency_ms_max"])
            if st["error"] and err is None:
                err = st.get("error_text") or "error %d" % st["error"]
            if not st["running"] and err is None:
                err = "pump thread ended"
        out.update({"rcf_pump_" + k: v for k, v in tot.items()})
        out["rcf_pump_latency_ms_p99"], out["rcf_pump_latency_ms_max"] = worst_p99, worst_max
        out["rcf_pump_samples_in"] = samples
        out["rcf_pump_subscriptions"] = len(self.subs)
        # of the late wake-ups, how much the thread spent runnable without a CPU (the host's scheduler, not the GPU)
        out["rcf_pump_late_wakeups_ms"] = sum(cl.pump.stats()["slow_sleep_ms_total"] for cl in self.classes.values() if cl.pump)
        out["rcf_pump_late_wakeups_on_run_queue_ms"] = sum(max(0.0, cl.pump.stats()["runq_ms_in_slow_sleeps"]) for cl in self.classes.values() if cl.pump)
        if err is not None:
            out["rcf_pump_error"] = err
        return out


class _FileFeeder:
    """a capture file into a counter-fed ring at samp_rate: rcf.sources.PacedSource's pacing, delivering to the ring"""

    def __init__(self, source_id, cfg, writer, block, per):
        class _Sink:                       # what PacedSource calls tb
            def __init__(s):
                s.sources = {}

            def feed(s, sid, blk):
                writer.write(blk)

            def feed_raw(s, sid, blk, fmt, scale, offset=0.0):
                writer.write(blk)
        cfg = dict(cfg)
        cfg["block_ms"] = block / float(cfg["samp_rate"]) * 1e3
        self.src = sources.PacedSource(_Sink(), source_id, cfg, pinned=False)

    def start(self):
        self.src.start()

    def stop(self):
        self.src.stop()
