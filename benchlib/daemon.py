"""The product process's own data path, measured: rcf.receiver + rcf.dataplane.NativeDataPlane (what `python -m rcf.frontend`
assembles) -- not the bench's own pump harness."""
import time


class _Sink:
    """a PUB socket without subscribers: counts what the egress thread hands over"""
    made = []

    def __init__(self, port):
        self.n = 0
        _Sink.made.append(self)

    def send(self, payload):
        self.n += len(payload)

    def close(self):
        pass


def daemon_leg(device, n_sources=32, channels_each=64, seconds=4.0):
    """`n_sources` x 20 Msps u8 sources in ONE receiver (rc_frontend/receiver.py:67-70: all configured sources in one
    process), `channels_each` reference-shaped 12.5 kHz channels requested on each through receiver.connect_channel and
    every one of them delivered to its socket by the egress thread: the daemon's path end to end (source rings ->
    native pump -> per-channel host rings -> socket.send), `seconds` of it."""
    from rcf import dataplane, receiver as receiver_mod

    class Cfg:
        receiver_split2 = False
        frontend_mode = "xlat"
        sources = {i: {"type": "synthetic", "center_freq": 400000000 + 25000000 * i, "samp_rate": 20000000, "seed": 50 + i,
                       "tile_samples": 1 << 21, "wire": "u8", "block_ms": 20.0, "carriers": []} for i in range(n_sources)}

    _Sink.made = []
    tb = receiver_mod.receiver(Cfg(), device=device)
    n_ch = n_sources * channels_each
    plane = dataplane.NativeDataPlane(tb, socket_factory=_Sink, period=0.02, max_channels=n_ch, out_ring_samples=1 << 13)
    try:
        t0 = time.perf_counter()
        for i in range(n_sources):
            for k in range(channels_each):
                tb.connect_channel(12500, 400000000 + 25000000 * i + (k - channels_each // 2) * 125000 + 12500)
        open_ms = (time.perf_counter() - t0) * 1e3 / n_ch
        cl, = plane.classes.values()
        plane.start()
        t_end = time.time() + 30
        while plane.stats()["rcf_pump_subscriptions"] < n_ch and time.time() < t_end:
            time.sleep(0.05)
        time.sleep(1.5)                                   # past the pump's warm-up second
        s0, b0, t0 = plane.stats(), [k.n for k in _Sink.made], time.time()
        time.sleep(seconds)
        s1, b1, wall = plane.stats(), [k.n for k in _Sink.made], time.time() - t0
        rates = [(b - a) / 8.0 / wall for a, b in zip(b0, b1) if b > 0]
        pump = next(iter(plane.detail().values()))
        return {
            "what": "rcf.receiver + rcf.dataplane.NativeDataPlane (the daemon's data path), %d x 20 Msps u8 sources, %d "
                    "channels each, all subscribed, %.1f s" % (n_sources, channels_each, wall),
            "sources": n_sources, "channels": n_ch, "block_ms": cl.block / cl.fs * 1e3, "seconds": wall,
            "input_Msps": (s1["rcf_pump_samples_in"] - s0["rcf_pump_samples_in"]) / wall / 1e6,
            "blocks": s1["rcf_pump_blocks_done"] - s0["rcf_pump_blocks_done"],
            "late_blocks": s1["rcf_pump_late"] - s0["rcf_pump_late"], "overruns": s1["rcf_pump_overruns"] - s0["rcf_pump_overruns"],
            "latency_ms_p99": s1["rcf_pump_latency_ms_p99"], "latency_ms_max": s1["rcf_pump_latency_ms_max"],
            "channels_delivering": len(rates), "channel_rate_min_sps": min(rates) if rates else 0.0,
            "channel_rate_max_sps": max(rates) if rates else 0.0,
            "egress_MBps": sum(b - a for a, b in zip(b0, b1)) / wall / 1e6, "egress_errors": plane.errors,
            "connect_channel_ms": open_ms,
            "late_wakeups_ms": s1["rcf_pump_late_wakeups_ms"] - s0["rcf_pump_late_wakeups_ms"],
            "late_wakeups_on_run_queue_ms": s1["rcf_pump_late_wakeups_on_run_queue_ms"] - s0["rcf_pump_late_wakeups_on_run_queue_ms"],
            "pump_stats": {k: pump[k] for k in ("group_blocks", "max_batch", "host_plan_ms", "host_wait_ms", "max_plan_ms",
                                                "max_wait_ms", "slow_waits", "slow_sleeps")},
        }
    finally:
        plane.stop()
        tb.close()
