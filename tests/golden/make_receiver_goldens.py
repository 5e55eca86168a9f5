#!/usr/bin/env python3
"""Generate tests/golden/receiver.json by RUNNING the reference's own receiver class.

rc_frontend/receiver.py's `receiver(gr.top_block)` is importable once GNU Radio, UHD, ZeroMQ and the publisher are
replaced by stand-ins (build container only: needs /root/reference): `gr.top_block` becomes an empty class, every block
constructor a MagicMock, `config.rc_config` a table of two or three USRP sources.  The control-plane methods the
device path mirrors -- connect_channel (source choice, offset, idle re-use), release_channel, source_offset (drift
report -> Hz -> retune) -- then run AS THEY STAND on seeded random sessions.  What is written is DATA: the calls, what
they returned or raised, and after every call the table of channels (source, rate, offset, in_use, port), the
accumulated offsets and the centre frequencies the SDR blocks were told to tune to.  No reference source travels.
"""
import json
import os
import random
import sys
import types
from unittest import mock

REF = "/root/reference/rc_frontend"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "receiver.json")


class _TopBlock:
    def __init__(self, *a, **k): pass
    def connect(self, *a, **k): pass
    def disconnect(self, *a, **k): pass
    def start(self, *a, **k): pass
    def stop(self, *a, **k): pass
    def wait(self, *a, **k): pass
    def lock(self, *a, **k): pass
    def unlock(self, *a, **k): pass


gr = mock.MagicMock()
gr.top_block = _TopBlock
gr.sizeof_gr_complex = 8
gnuradio = types.ModuleType("gnuradio")
gnuradio.gr = gr
for name in ("filter", "blocks", "zeromq", "uhd", "analog"):
    setattr(gnuradio, name, mock.MagicMock())
sys.modules["gnuradio"] = gnuradio
sys.modules["gnuradio.gr"] = gr
for name in ("filter", "blocks", "zeromq", "uhd", "analog"):
    sys.modules["gnuradio." + name] = getattr(gnuradio, name)
sys.modules["gnuradio.filter.pfb"] = mock.MagicMock()
sys.modules["gnuradio.filter.firdes"] = mock.MagicMock()
sys.modules["gnuradio.filter.optfir"] = mock.MagicMock()
gnuradio.filter.pfb = sys.modules["gnuradio.filter.pfb"]
gnuradio.filter.firdes = sys.modules["gnuradio.filter.firdes"]
gnuradio.filter.optfir = sys.modules["gnuradio.filter.optfir"]
gnuradio.uhd.usrp_source.side_effect = lambda *a, **k: mock.MagicMock()      # one SDR block per source, not one shared mock
sys.modules["osmosdr"] = mock.MagicMock()
sys.modules["zmq"] = mock.MagicMock()
pubmod = types.ModuleType("redis_channel_publisher")
pubmod.redis_channel_publisher = lambda **k: None
sys.modules["redis_channel_publisher"] = pubmod

CONFIGS = {
    "two_sources": dict(split2=False, sources={
        0: dict(type="usrp", device_addr="a", otw_format="sc16", args="", samp_rate=2400000, center_freq=855050000, rf_gain=10),
        1: dict(type="usrp", device_addr="b", otw_format="sc16", args="", samp_rate=2400000, center_freq=857000000, rf_gain=10)}),
    "three_sources_one_with_offset": dict(split2=False, sources={
        0: dict(type="usrp", device_addr="a", otw_format="sc16", args="", samp_rate=8000000, center_freq=854000000, rf_gain=10, offset=1200),
        1: dict(type="usrp", device_addr="b", otw_format="sc16", args="", samp_rate=2400000, center_freq=857000000, rf_gain=10),
        2: dict(type="usrp", device_addr="c", otw_format="sc16", args="", samp_rate=10000000, center_freq=862000000, rf_gain=10)}),
    "scan_mode": dict(split2=False, scan_mode=True, sources={
        0: dict(type="usrp", device_addr="a", otw_format="sc16", args="", samp_rate=2400000, center_freq=855050000, rf_gain=10)}),
    "split2": dict(split2=True, sources={
        0: dict(type="usrp", device_addr="a", otw_format="sc16", args="", samp_rate=8000000, center_freq=855000000, rf_gain=10)}),
}
FREQS = [855050000, 854987500, 855500000, 856100000, 857000000, 857900000, 858300000, 853000000, 851012500, 860000000,
         866900000, 900000000, 854000000, 850100000, 5000, 857012500]
OFFSETS = [0.1, 0.25, 0.6, 0.9, 1.0, 1.2, 1.5, 2.0, -0.3, -0.75, -1.01, -2.5, 3.0, -4.0, 0.0, 130.0]

sys.path.insert(0, REF)
golden = {"configs": {k: {"split2": v["split2"], "scan_mode": bool(v.get("scan_mode")), "sources": {str(i): s for i, s in v["sources"].items()}} for k, v in CONFIGS.items()},
          "sessions": []}


def snapshot(tb, order):
    chans = []
    for bid in order:
        if bid in tb.channels:
            c = tb.channels[bid]
            chans.append({"n": order.index(bid), "source_id": c.source_id, "channel_rate": c.channel_rate, "offset": c.offset,
                          "in_use": c.in_use, "port": c.port, "idle": c.channel_close_time != 0})
    return {"channels": chans,
            "accumulated": {str(i): tb.sources[i].get("accumulated_offset") for i in tb.sources},
            "sources": {str(i): [tb.sources[i]["center_freq"], tb.sources[i]["samp_rate"]] for i in tb.sources}}


def session(cfg_name, seed):
    import copy
    cfg = CONFIGS[cfg_name]
    config = types.ModuleType("config")

    class rc_config:
        def __init__(self):
            self.sources = copy.deepcopy(cfg["sources"])
            self.frontend_mode = "xlat"
            self.receiver_split2 = cfg["split2"]
            if cfg.get("scan_mode"):
                self.scan_mode = True
    config.rc_config = rc_config
    sys.modules["config"] = config
    for m in ("receiver", "channel"):
        sys.modules.pop(m, None)
    import receiver as R
    rs = random.Random(seed)
    random.seed(seed)                                # the reference draws its egress ports from the module-level generator
    tb = R.receiver(None)
    order, steps = [], []
    tuned = {}
    for i in tb.sources:                             # every SDR block records what it is told to tune to
        blk = tb.sources[i]["block"]
        blk.set_center_freq = (lambda i: lambda f, ch=0: tuned.setdefault(i, []).append(f))(i)
    for _ in range(rs.randint(4, 30)):
        u = rs.random()
        tuned.clear()
        if u < 0.5 or not order:
            call = ["connect_channel", rs.choice([12500, 12500, 25000]),
                    rs.choice(FREQS if not cfg.get("scan_mode") else [5000, -250000, 100, 600000, 855050000, 1100000, 0])]
        elif u < 0.8:
            call = ["release_channel", rs.choice(list(range(len(order))) + [-1])]
        else:
            call = ["source_offset", rs.choice(list(range(len(order))) + [-1]), rs.choice(OFFSETS)]
        design = None
        try:
            if call[0] == "connect_channel":
                gnuradio.filter.freq_xlating_fir_filter_ccc.reset_mock()
                gnuradio.filter.firdes.low_pass_2.reset_mock()
                bid, port = tb.connect_channel(call[1], call[2])
                if bid not in order and bid is not False:
                    order.append(bid)
                    # a NEW channel: what rc_frontend/channel.py:31-35 asked GNU Radio for (decimation, filter design
                    # arguments, xlating offset and rate), read off the stand-ins
                    xl = gnuradio.filter.freq_xlating_fir_filter_ccc.call_args[0]
                    lp = gnuradio.filter.firdes.low_pass_2.call_args[0]
                    design = {"decim": xl[0], "offset": xl[2], "samp_rate": xl[3], "low_pass_2": [float(v) for v in lp[:5]],
                              "window_is_hamming": lp[5] is gnuradio.filter.firdes.WIN_HAMMING}
                ret = [order.index(bid) if bid in order else bid, port]
            elif call[0] == "release_channel":
                ret = tb.release_channel(order[call[1]] if call[1] >= 0 else "no-such-id")
            else:
                ret = tb.source_offset(order[call[1]] if call[1] >= 0 else "no-such-id", call[2])
            exc = None
        except Exception as e:
            ret, exc = None, "%s: %s" % (type(e).__name__, e)
        steps.append({"call": call, "returns": ret, "raises": exc, "design": design, "tuned": {str(k): list(v) for k, v in tuned.items()},
                      "after": snapshot(tb, order)})
    return {"config": cfg_name, "seed": seed, "steps": steps}


for name in CONFIGS:
    for seed in range(24):
        golden["sessions"].append(session(name, seed))
with open(OUT, "w") as f:
    json.dump(golden, f, indent=0, sort_keys=True)
print("wrote", OUT, len(golden["sessions"]), "sessions,", sum(len(s["steps"]) for s in golden["sessions"]), "calls")
os._exit(0)
