#!/usr/bin/env python3
"""In-process probe of the native data plane: one synthetic source, one channel, everything the egress thread hands to the
channel's socket collected and compared with the oracle; prints where the streams differ."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
import numpy as np
from oracle import cbind as OC, grspec as G
from rcf import dataplane, receiver as receiver_mod, sources

FS, CR, F_OFF = 2400000, 12500, -62500
WIRE = os.environ.get("WIRE", "cf32")
NSRC = int(os.environ.get("NSRC", 1))


class Cfg:
    receiver_split2 = False
    frontend_mode = "xlat"
    sources = {i: {"type": "synthetic", "center_freq": 855050000 + 3000000 * i, "samp_rate": FS, "seed": 1001 + i, "tile_samples": 1 << 20,
                   "wire": (WIRE if WIRE != "mixed" else ("u8" if i % 2 else "cf32")), "block_ms": 20.0, "carriers": [{"f_off": F_OFF, "f_mod": 1000.0, "dev": 2500.0, "snr_db": 30.0}]}
               for i in range(NSRC)}


class Sink:
    made = []

    def __init__(self, port):
        self.chunks = []
        Sink.made.append(self)

    def send(self, b):
        self.chunks.append(b)

    def close(self):
        pass


tb = receiver_mod.receiver(Cfg(), device=0)
plane = dataplane.NativeDataPlane(tb, socket_factory=Sink, period=0.01)
plane.start()
if os.environ.get("KM"):
    import threading
    tb.enable_kernel_metrics(32)
    def poll():
        while plane.continue_running:
            plane.stats(); tb.metrics(); time.sleep(float(os.environ.get("KM")))
    threading.Thread(target=poll, daemon=True).start()
time.sleep(0.3)
ids = [tb.connect_channel(CR, 855050000 + 3000000 * i + F_OFF)[0] for i in range(NSRC)]
time.sleep(float(os.environ.get("SECS", 2.0)))
starts = {b: tb.channels[b].start_sample for b in ids}
st = plane.stats()
for cl in plane.classes.values():
    print(cl.key, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in cl.pump.stats().items()})
plane.stop()
print({k: v for k, v in st.items()})
D, taps = G.channel_params(FS, CR)
for i, b in enumerate(ids):
    sock = plane_sock = None
    got = np.frombuffer(b"".join(Sink.made[i].chunks), dtype=np.complex64)
    src = dict(Cfg.sources[i])
    tile = sources.synthetic_tile(src)
    if src["wire"] != "cf32":
        scale, off = sources.WIRE_SCALE[src["wire"]]
        raw = sources.to_wire(tile, src["wire"]).astype(np.float32)
        tile = ((raw - np.float32(off)) * np.float32(scale)).view(np.complex64)
    start = -(-starts[b] // D) * D
    need = (len(got) + 2000) * D
    reps = (start % len(tile) + need) // len(tile) + 2
    x = np.tile(tile, reps)[start % len(tile):][:need]
    ct, incr = OC.xlating_composite(taps, D, float(F_OFF), float(FS))
    want, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[1.0])
    want = want[0]
    probe = 64
    win = np.lib.stride_tricks.sliding_window_view(want[: 2000 + probe], probe)
    k0 = int(np.argmin(np.abs(win - got[100:100 + probe]).sum(axis=1))) - 100
    n = min(len(got), len(want) - max(k0, 0))
    ref = want[k0:k0 + n] if k0 >= 0 else want[:n]
    g = got[:n] if k0 >= 0 else got[-k0:-k0 + n]
    m = min(len(ref), len(g))
    d = np.abs(g[:m] - ref[:m])
    bad = np.nonzero(d > 1e-3 * np.abs(ref[:m]).mean())[0]
    print("source %d: %d samples, start %d, k0 %d, rel rms %.2e, bad samples %d at %s (chunk sizes %s...)" % (
        i, len(got), start, k0, float(np.sqrt(np.mean(d ** 2) / np.mean(np.abs(ref[:m]) ** 2))), len(bad), bad[:12],
        [len(c) // 8 for c in Sink.made[i].chunks[:6]]))
tb.close()
