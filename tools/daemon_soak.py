#!/usr/bin/env python3
"""Soak of the product's data plane under channel churn: SOURCES x 20 Msps u8 sources on the native pump, CHANNELS channels
held at any time, and every CHURN_MS one of them released and swept and another requested somewhere else -- subscriptions
come and go under the running pump for SECONDS.  Prints one JSON object (late blocks, errors, channels delivering)."""
import json, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "radiocapture-rf_amd")]
from rcf import dataplane, receiver as receiver_mod

S = int(os.environ.get("SOURCES", 8)); C = int(os.environ.get("CHANNELS", 256)); T = float(os.environ.get("SECONDS", 30))
churn = float(os.environ.get("CHURN_MS", 20)) * 1e-3


class Cfg:
    receiver_split2 = False
    frontend_mode = "xlat"
    sources = {i: {"type": "synthetic", "center_freq": 400000000 + 25000000 * i, "samp_rate": 20000000, "seed": 70 + i,
                   "tile_samples": 1 << 21, "wire": "u8", "block_ms": 20.0, "carriers": []} for i in range(S)}


class Sink:
    n = 0
    made = 0

    def __init__(self, port):
        Sink.made += 1

    def send(self, b):
        Sink.n += len(b)

    def close(self):
        pass


rng = random.Random(5)
tb = receiver_mod.receiver(Cfg(), device=0)
tb.channel_idle_timeout = 0.05
plane = dataplane.NativeDataPlane(tb, socket_factory=Sink, period=0.01, max_channels=C + 64)


def request():
    i = rng.randrange(S)
    return tb.connect_channel(12500, 400000000 + 25000000 * i + rng.randrange(-700, 700) * 12500)[0]


held = [request() for _ in range(C)]
plane.start()
time.sleep(2.0)
s0, n0, t0 = plane.stats(), Sink.n, time.time()
swaps = 0
while time.time() - t0 < T:
    k = rng.randrange(len(held))
    tb.release_channel(held[k])
    held[k] = request()                      # (usually re-uses the idle channel object with a new offset; sometimes a new one)
    if swaps % 16 == 0:
        tb.sweep_idle_channels()
    swaps += 1
    time.sleep(churn)
s1, wall = plane.stats(), time.time() - t0
out = {"sources": S, "channels_held": C, "seconds": wall, "swaps": swaps, "sockets_made": Sink.made,
       "blocks": s1["rcf_pump_blocks_done"] - s0["rcf_pump_blocks_done"],
       "late": s1["rcf_pump_late"] - s0["rcf_pump_late"], "overruns": s1["rcf_pump_overruns"] - s0["rcf_pump_overruns"],
       "latency_ms_p99": s1["rcf_pump_latency_ms_p99"], "latency_ms_max": s1["rcf_pump_latency_ms_max"],
       "subscriptions_at_end": s1["rcf_pump_subscriptions"], "channels_open_at_end": len(tb.channels),
       "egress_MBps": (Sink.n - n0) / wall / 1e6, "expected_MBps": C * 25000 * 8 / 1e6,
       "egress_errors": plane.errors, "pump_error": s1.get("rcf_pump_error"), "healthy": tb.healthy(),
       "late_wakeups_ms": s1["rcf_pump_late_wakeups_ms"] - s0["rcf_pump_late_wakeups_ms"],
       "of_them_on_a_run_queue_ms": s1["rcf_pump_late_wakeups_on_run_queue_ms"] - s0["rcf_pump_late_wakeups_on_run_queue_ms"]}
plane.stop()
tb.close()
print(json.dumps(out))
