"""Paced sources: the `'type'` values this build adds next to the reference's `'rtlsdr' | 'bladerf' | 'usrp'`
(/root/reference/rc_frontend/receiver.py:74-191 builds an osmosdr / UHD source block per configured source and lets
the hardware clock pace the flowgraph).  Hardware drivers are out of scope here; what stands in for them:

  'synthetic'   a seeded wideband stream (unit noise + NBFM carriers), generated once as a periodic tile and replayed
                at exactly `samp_rate` of wall-clock time.  Keys: 'carriers' (list of {f_off, f_mod, dev, snr_db}),
                'seed', 'tile_samples', 'wire' ('cf32' | 'u8' | 's8' | 's16': what an rtlsdr / USRP sc8 / sc16 link
                would deliver; converted on the GPU by rcf_push_raw)
  'file'        a capture file replayed at `samp_rate`: 'path', 'format' ('cf32' as file_to_wav.py / log_dat write,
                'u8' as rtl_sdr writes, 's8', 's16'), 'loop' (default True)

A PacedSource is one thread per source: every `block_ms` it hands the next block -- staged in one of two pinned host
buffers (rcf_host_alloc), so the copy of block n+1 overlaps the kernels of block n -- to receiver.feed /
receiver.feed_raw, at t0 + (k + 1) * block / samp_rate -- the instant the block's last sample exists.  It never runs
ahead of the clock, and when it falls behind it does not drop samples: a block whose delivery starts more than one block
period after that instant is counted as `late` and the source catches up (a real SDR would have overrun its ring instead).
"""
from __future__ import annotations

import math
import threading
import time

import numpy as np

from . import synth

WIRE = {"cf32": (np.complex64, None), "u8": (np.uint8, 1), "s8": (np.int8, 2), "s16": (np.int16, 3)}   # native.FMT_*
# wire scale / offset so that the converted stream has the level of the cf32 one (rcf_push_raw: (v - offset) * scale)
WIRE_SCALE = {"u8": (1.0 / 32.0, 127.4), "s8": (1.0 / 32.0, 0.0), "s16": (1.0 / 8192.0, 0.0)}


def to_wire(x: np.ndarray, wire: str) -> np.ndarray:
    """cf32 -> the interleaved integer samples an SDR of that wire format would have sent for it"""
    if wire == "cf32":
        return np.ascontiguousarray(x, dtype=np.complex64)
    scale, offset = WIRE_SCALE[wire]
    dt, _ = WIRE[wire]
    info = np.iinfo(dt)
    v = np.round(x.view(np.float32) / scale + offset)
    return np.clip(v, info.min, info.max).astype(dt)


def synthetic_tile(src: dict) -> np.ndarray:
    fs = float(src["samp_rate"])
    n = int(src.get("tile_samples", 1 << 20))
    rng = np.random.Generator(np.random.PCG64(int(src.get("seed", 1))))
    x = synth.awgn(rng, n).astype(np.complex128)
    for c in src.get("carriers", []):
        # whole cycles of carrier and tone per tile, so that the replayed tile has no phase jump at the seam
        f_off = round(float(c["f_off"]) * n / fs) * fs / n
        f_mod = max(1, round(float(c.get("f_mod", 1000.0)) * n / fs)) * fs / n
        x += synth.nbfm_carrier(n, fs, f_off, f_mod, float(c.get("dev", 2500.0)),
                                synth.snr_amp(float(c.get("snr_db", 30.0)), 12500.0, fs))
    return x.astype(np.complex64)


class PacedSource:
    def __init__(self, tb, source_id, src: dict, block_ms=20.0, pinned=True, clock=time.perf_counter, sleep=time.sleep):
        """tb: rcf.receiver.receiver; source_id: key of tb.sources the stream belongs to; src: the config's source dict"""
        self.tb, self.source_id, self.src = tb, source_id, src
        self.fs = float(src["samp_rate"])
        self.kind = src["type"]
        self.wire = src.get("wire" if self.kind == "synthetic" else "format", "cf32")
        if self.wire not in WIRE:
            raise ValueError("source %s: wire format %r" % (source_id, self.wire))
        self.block = max(1, int(round(self.fs * float(src.get("block_ms", block_ms)) * 1e-3)))
        self.loop = bool(src.get("loop", True))
        self.clock, self.sleep = clock, sleep
        self.blocks = self.late = self.samples = 0
        self.max_lag_s = 0.0
        self.feed_s = []                                    # wall time of each feed() call (latency of the hand-over)
        self.continue_running = True
        self.error = None
        self._thread = None
        dt, _ = WIRE[self.wire]
        per = 1 if self.wire == "cf32" else 2               # array elements per sample
        if self.kind == "synthetic":
            self._data = to_wire(synthetic_tile(src), self.wire)
        elif self.kind == "file":
            self._data = np.memmap(src["path"], dtype=dt, mode="r")
            if len(self._data) < per:
                raise ValueError("source %s: %s is empty" % (source_id, src["path"]))
        else:
            raise ValueError("source %s: no driver for type %r in this build (types: 'synthetic', 'file'; SDR hardware "
                             "is delivered through receiver.feed by whatever owns the device)" % (source_id, self.kind))
        self._per = per
        self._n = len(self._data) // per
        self._at = 0
        self._pin = None
        self._stage = [np.empty(self.block * per, dtype=dt) for _ in range(2)]
        if pinned:
            try:
                from . import native
                self._pin = [native.PinnedArray(self.block * per, dt) for _ in range(2)]
                self._stage = [p.array for p in self._pin]
            except Exception:
                self._pin = None                            # no librcf / no device (tests with a stub front-end)

    def next_block(self, k):
        """samples [at, at + block) of the (looped) stream into staging buffer k % 2; None at the end of a file"""
        buf, per, n = self._stage[k & 1], self._per, self.block
        got = 0
        while got < n:
            if self._at >= self._n:
                if not self.loop:
                    break
                self._at = 0
            take = min(n - got, self._n - self._at)
            buf[got * per:(got + take) * per] = self._data[self._at * per:(self._at + take) * per]
            self._at += take
            got += take
        if got == 0:
            return None
        return buf[: got * per]

    def deliver(self, blk):
        if self.wire == "cf32":
            self.tb.feed(self.source_id, blk)
        else:
            scale, offset = WIRE_SCALE[self.wire]
            self.tb.feed_raw(self.source_id, blk, WIRE[self.wire][1], scale, offset)

    def run(self, max_blocks=None):
        t0 = self.clock()
        k = 0
        try:
            while self.continue_running and (max_blocks is None or k < max_blocks):
                blk = self.next_block(k)
                if blk is None:
                    break
                # block k is delivered once its LAST sample exists -- (k + 1) block periods after the start: a source
                # never hands over samples that a real SDR would not have produced yet -- and it is late when the
                # delivery starts more than a block period after that (the next block is then already complete)
                due = t0 + (k + 1) * self.block / self.fs
                now = self.clock()
                if now < due:
                    self.sleep(due - now)
                else:
                    lag = now - due
                    self.max_lag_s = max(self.max_lag_s, lag)
                    if lag > self.block / self.fs:
                        self.late += 1
                t1 = self.clock()
                self.deliver(blk)
                self.feed_s.append(self.clock() - t1)
                if len(self.feed_s) > 4096:
                    del self.feed_s[:2048]
                self.blocks += 1
                self.samples += len(blk) // self._per
                k += 1
        except Exception as e:                              # receiver.feed has already marked the receiver unhealthy
            self.error = "%s: %s" % (type(e).__name__, e)
        return self

    def start(self):
        self._thread = threading.Thread(target=self.run, name="source-%s" % self.source_id, daemon=True)
        self._thread.start()
        return self

    def stop(self):
        self.continue_running = False
        if self._thread is not None:
            self._thread.join(timeout=5)
        if self._pin and not (self._thread is not None and self._thread.is_alive()):   # (a push still in flight keeps its buffers)
            for p in self._pin:
                p.free()
            self._pin = None

    def stats(self):
        f = sorted(self.feed_s)
        return {"blocks": self.blocks, "samples": self.samples, "late_blocks": self.late,
                "max_lag_ms": self.max_lag_s * 1e3, "block_ms": self.block / self.fs * 1e3,
                "feed_ms_p50": f[len(f) // 2] * 1e3 if f else None,
                "feed_ms_p99": f[min(len(f) - 1, int(math.ceil(len(f) * 0.99)) - 1)] * 1e3 if f else None,
                "error": self.error}


def start_paced_sources(tb, **kw):
    """one PacedSource per source of the receiver whose config type is 'synthetic' or 'file' (receiver_split2 halves
    share their parent's stream: one thread per REAL source).  Others are left to feed()."""
    started, seen = [], set()
    for sid in sorted(tb.sources):
        real = tb.sources[sid]["source_id"]
        cfg = tb.realsources[real]
        if real in seen or cfg.get("type") not in ("synthetic", "file"):
            continue
        seen.add(real)
        started.append(PacedSource(tb, sid, cfg, **kw).start())
    return started
