"""Host-side mirror of the reference's control plane, checked against goldens captured from the
reference itself (tests/golden/protocol.json, made by tests/golden/make_protocol_goldens.py)."""
import json
import math
import os
import random
import types

import numpy as np

import pytest

from rcf import frontend_connector as FC
from rcf import protocol, receiver, registry

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "protocol.json")))


class ScriptedSocket:
    def __init__(self, replies, log):
        self.replies, self.log = replies, log

    def send_string(self, s):
        self.log.append(s)

    def recv_string(self):
        return self.replies.pop(0)

    def close(self):
        pass


class FakeRCM:
    def get_channelizer_for_frequency(self, f):
        return ("10.0.0.5", 4242)


@pytest.mark.parametrize("case", GOLD["connector"], ids=lambda c: c["name"])
def test_connector_emits_reference_wire_strings(case):
    sent, replies, connects = [], [], []

    def factory(host, port):
        connects.append("tcp://%s:%s" % (host, port))
        return ScriptedSocket(replies, sent)

    fc = FC.frontend_connector("parent-uuid", FakeRCM(), transport_factory=factory, heartbeat=False)
    for step in case["steps"]:
        del sent[:], connects[:]
        replies[:] = list(step["replies"])
        ret = getattr(fc, step["call"])(*step["args"])
        ret = list(ret) if isinstance(ret, tuple) else ret
        assert sent == step["requests"], step
        assert connects == step["connects"]
        assert ret == step["returns"], step          # incl. port as STRING, (False, False) on failure
        assert fc.host == step["host"]


@pytest.mark.parametrize("q", GOLD["rcm"], ids=lambda q: "%s-%s" % (q["table"], q["frequency"]))
def test_channelizer_selection_rule(q):
    mgr = registry.redis_channelizer_manager(clients=[], start_thread=False)
    mgr.channelizers = q["channelizers"]
    random.seed(0)
    assert list(mgr.get_channelizer_for_frequency(q["frequency"])) == q["result"]


# ------------------------------------------------------------------ server side, with a stub front-end
class StubFrontend:
    """Stands in for native.Frontend on GPU-less boxes: records calls, mimics the C ABI's checks."""
    instances = []

    def __init__(self, samp_rate, center_freq, device=0):
        self.samp_rate, self.center_freq = samp_rate, center_freq
        self.chans, self.next, self.shift, self.closed = {}, 1, 0.0, False
        StubFrontend.instances.append(self)

    def chan_open(self, cr, off):
        q = int(self.samp_rate / cr)
        if q < 2 or q % 2 or not abs(off) < self.samp_rate / 2:
            raise RuntimeError("librcf error -8")
        cid = self.next
        self.next += 1
        self.chans[cid] = dict(cr=cr, off=off, D=q // 2)
        return cid

    def chan_open_taps(self, src, decim, taps, off):
        cid = self.next
        self.next += 1
        self.chans[cid] = dict(cr=None, off=off, D=int(decim), src=src, ntaps=len(taps))
        return cid

    def chan_info(self, cid):
        c = self.chans[cid]
        return dict(decim=c["D"], ntaps=c.get("ntaps", 1), out_rate=self.samp_rate / c["D"], offset_hz=c["off"])

    def chan_set_offset(self, cid, off):
        self.chans[cid]["off"] = off

    def chan_close(self, cid):
        del self.chans[cid]

    def source_shift(self, d):
        self.shift += d

    def close(self):
        self.closed = True


def make_receiver(scan_mode=False):
    cfg = types.SimpleNamespace(
        sources={0: dict(type="synthetic", center_freq=855050000, samp_rate=2400000),
                 1: dict(type="synthetic", center_freq=857000000, samp_rate=2400000)},
        frontend_mode="xlat", scan_mode=scan_mode)
    StubFrontend.instances = []
    return receiver.receiver(cfg, frontend_factory=StubFrontend)


def test_server_protocol_roundtrip_with_own_connector():
    tb = make_receiver()
    now = [1000.0]
    srv = protocol.FrontendServer(tb, clock=lambda: now[0])
    rcm = FakeRCM()
    fc = FC.frontend_connector("p", rcm, transport_factory=lambda h, p: protocol.LoopbackTransport(srv),
                               heartbeat=False)
    cid, port = fc.create_channel(12500, 854987500)
    assert cid in tb.channels and isinstance(port, str) and 10000 <= int(port) <= 60000
    ch = tb.channels[cid]
    assert ch.in_use and ch.source_id == 0 and ch.offset == 854987500 - 855050000 and ch.decim == 96
    assert fc.heartbeat_once() is True
    assert fc.report_offset(0.25) is True               # 1 Hz: inside the +-5 Hz dead band
    assert StubFrontend.instances[0].shift == 0.0
    assert fc.report_offset(2.0) is True                # > 1 -> x50 = 100 Hz
    assert StubFrontend.instances[0].shift == 100.0
    assert tb.sources[0]["accumulated_offset"] == 100.0
    assert fc.release_channel() == cid
    assert not tb.channels[cid].in_use and tb.channels[cid].channel_close_time > 0
    # idle channel of the same (source, rate) is re-used and retuned (receiver.py:311-319)
    cid2, _ = fc.create_channel(12500, 855500000)
    assert cid2 == cid and tb.channels[cid].offset == 450000 and tb.channels[cid].in_use
    # nearest source wins (receiver.py:288-291)
    cid3, _ = fc.create_channel(12500, 856500000)
    assert tb.channels[cid3].source_id == 1
    # out of band -> 'na' -> (False, False)
    assert fc.create_channel(12500, 900000000) == (False, False)
    # non-integral decimation is refused by the ABI -> 'na'
    assert fc.create_channel(800000, 855000000) == (False, False)   # int(fs/cr) = 3


def test_server_edge_cases_follow_reference_handler():
    tb = make_receiver()
    now = [0.0]
    srv = protocol.FrontendServer(tb, clock=lambda: now[0])
    assert srv.handle("hb,5") == "fail,5"
    assert srv.handle("hb,x") == "fail,0"
    # create before connect: channel is made, then released, 'na' (receiver.py:527-533)
    assert srv.handle("create,9,12500,855000000") == "na,855000000"
    assert len(tb.channels) == 1 and not list(tb.channels.values())[0].in_use
    assert srv.handle("connect") == "connect,0"
    assert srv.handle("connect") == "connect,1"
    r = srv.handle("create,0,12500,855000000").split(",")
    assert r[0] == "create" and r[1] in tb.channels
    assert srv.handle("release,0,%s" % r[1]) == "release,%s" % r[1]
    assert srv.handle("release,zz") == "na\n"
    assert srv.handle("release,0,not-a-channel") == "release,not-a-channel"     # release_channel -> True
    assert srv.handle("offset,0,%s,0.1" % r[1]) == "offset,0"
    assert srv.handle("scan_mode_set_freq,770000000") == "success"
    assert tb.sources[0]["center_freq"] == 770000000
    # heartbeat expiry after 5 s releases the client's channels (receiver.py:652-680)
    r2 = srv.handle("create,1,12500,857100000").split(",")
    assert tb.channels[r2[1]].in_use
    now[0] = 4.0
    assert srv.handle("hb,0") == "hb,0"
    now[0] = 6.0
    assert srv.tick() == [1]
    assert not tb.channels[r2[1]].in_use and 1 not in srv.clients and 0 in srv.clients
    assert srv.handle("quit,0") == "quit,0" and 0 not in srv.clients
    # idle sweep: 10 s idle timeout, checked at most every 20 s (receiver.py:51,635-648)
    n_before = len(tb.channels)
    now[0] = 6.0 + 25.0
    tb.last_channel_cleanup = 0.0
    for ch in tb.channels.values():
        if ch.channel_close_time:
            ch.channel_close_time = 6.0
    srv.last_status = 0.0
    srv.tick()
    assert len(tb.channels) < n_before
    assert all(c.in_use or now[0] - c.channel_close_time <= 10 for c in tb.channels.values())


def test_scan_mode_relative_frequency():
    tb = make_receiver(scan_mode=True)
    bid, _ = tb.connect_channel(12500, 25000)             # freq < 10 MHz: offset = freq (receiver.py:304)
    assert tb.channels[bid].offset == 25000 and tb.channels[bid].source_id == 0
    assert tb.source_offset(bid, 3.0) is False            # scan mode ignores drift reports


def test_registry_publish_and_expiry():
    class FakeRedis:
        def __init__(self): self.kv, self.sets = {}, {}
        def sadd(self, k, v): self.sets.setdefault(k, set()).add(v)
        def set(self, k, v): self.kv[k] = v
        def smembers(self, k): return set(self.sets.get(k, set()))
        def get(self, k): return self.kv.get(k)
        def srem(self, k, v): self.sets.get(k, set()).discard(v)
        def delete(self, k): self.kv.pop(k, None)
    r = FakeRedis()
    sources = {0: {"center_freq": 855050000, "samp_rate": 2400000}}
    pub = registry.redis_channel_publisher(sources=sources, channels={}, port=5555, index=3, client=r,
                                           address="10.1.2.3", start_thread=False,
                                           extra=lambda: {"msps_in": 2.4})
    d = pub.publish_once(now=100.0)
    for key in ("instance_uuid", "start_time", "current_time", "hostname", "pid", "address", "port",
                "channel_count", "source_count", "sources", "index"):
        assert key in d                                  # redis_channel_publisher.py:63-80 key set
    assert d["sources"] == [(855050000, 2400000)] and d["port"] == 5555 and d["msps_in"] == 2.4
    mgr = registry.redis_channelizer_manager(index=3, clients=[r], start_thread=False)
    mgr.poll_once(now=101.0)
    assert mgr.get_channelizer_for_frequency(855000000) == ("10.1.2.3", 5555)
    assert mgr.get_channelizer_for_frequency(900000000) == (None, None)
    mgr.poll_once(now=106.5)                             # > 5 s stale: expired and deleted
    assert mgr.channelizers == {} and r.smembers("channelizers") == set()
    other = registry.redis_channelizer_manager(index=4, clients=[r], start_thread=False)
    pub.publish_once(now=200.0)
    other.poll_once(now=200.5)
    assert other.channelizers == {}                      # index filter (redis_channelizer_manager.py:94)


def test_data_plane_fault_stops_the_heartbeat_and_metrics_are_additive():
    """SURVEY 5 (failure handling / metrics): a failed push (GPU / driver error) marks the receiver unhealthy, the
    registry publisher stops publishing (the manager expires the record after 5 s), and receiver.metrics() only adds
    keys to the reference's record."""
    class FakeRedis:
        def __init__(self): self.kv, self.sets = {}, {}
        def sadd(self, k, v): self.sets.setdefault(k, set()).add(v)
        def set(self, k, v): self.kv[k] = v
        def smembers(self, k): return set(self.sets.get(k, set()))
        def get(self, k): return self.kv.get(k)
        def srem(self, k, v): self.sets.get(k, set()).discard(v)
        def delete(self, k): self.kv.pop(k, None)

    rx = make_receiver()
    fe = StubFrontend.instances[0]
    pushed = []
    fe.push = lambda iq: pushed.append(len(iq))
    r = FakeRedis()
    pub = registry.redis_channel_publisher(sources=rx.sources, channels=rx.channels, port=5555, client=r,
                                           address="10.0.0.1", start_thread=False, extra=rx.metrics, health=rx.healthy)
    rx.feed(0, np.zeros(1000, np.complex64))
    block_id, _ = rx.connect_channel(12500, 855000000)
    d = pub.publish_once(now=10.0)
    assert d["rcf_samples_in"] == 1000 and d["rcf_healthy"] is True and d["rcf_channels_open"] == 1
    assert d["rcf_channels_in_use"] == 1 and "rcf_fault" not in d
    for key in ("instance_uuid", "start_time", "current_time", "hostname", "pid", "address", "port",
                "channel_count", "source_count", "sources"):
        assert key in d                                  # the reference's keys are all still there
    mgr = registry.redis_channelizer_manager(clients=[r], start_thread=False)
    mgr.poll_once(now=10.5)
    assert mgr.get_channelizer_for_frequency(855000000) == ("10.0.0.1", 5555)

    def broken(iq):
        raise RuntimeError("librcf error -2: hipErrorLaunchFailure")
    fe.push = broken
    with pytest.raises(RuntimeError):
        rx.feed(0, np.zeros(10, np.complex64))
    assert not rx.healthy() and "hipErrorLaunchFailure" in rx.fault
    assert pub.publish_once(now=11.0) is None            # no heartbeat
    assert json.loads(r.get(pub.instance_uuid))["current_time"] == 10.0
    mgr.poll_once(now=16.0)                              # > 5 s since the last record: gone
    assert mgr.get_channelizer_for_frequency(855000000) == (None, None)
    assert rx.metrics()["rcf_fault"].startswith("RuntimeError")
    # a health callable that itself fails counts as unhealthy
    pub2 = registry.redis_channel_publisher(sources=rx.sources, channels={}, port=1, client=r, start_thread=False,
                                            health=lambda: 1 / 0)
    assert pub2.publish_once() is None


def test_receiver_split2_makes_two_half_rate_sources_per_real_source():
    """receiver.py:205-237: centre -/+ fs/4, rate fs/2, through a /2 xlating FIR with firdes.low_pass taps;
    channels of a half are chained behind it with channel.py's rule at the HALF rate."""
    cfg = types.SimpleNamespace(
        sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=2400000)},
        frontend_mode="xlat", scan_mode=False, receiver_split2=True)
    StubFrontend.instances = []
    tb = receiver.receiver(cfg, frontend_factory=StubFrontend)
    assert len(StubFrontend.instances) == 1 and len(tb.sources) == 2
    fe = StubFrontend.instances[0]
    lo, hi = tb.sources[0], tb.sources[1]
    assert (lo["center_freq"], lo["samp_rate"]) == (855000000 - 600000, 1200000)
    assert (hi["center_freq"], hi["samp_rate"]) == (855000000 + 600000, 1200000)
    halves = [fe.chans[lo["parent_chan"]], fe.chans[hi["parent_chan"]]]
    assert [h["off"] for h in halves] == [-600000.0, 600000.0]
    assert all(h["D"] == 2 and h["src"] == -1 for h in halves)
    # firdes.low_pass(1, fs, fs/4, fs/8): int(53 fs / (22 fs/8)) = 19 taps (odd already)
    assert halves[0]["ntaps"] == 19
    bid, port = tb.connect_channel(12500, 855000000 + 500000)      # nearest centre: the upper half
    ch = tb.channels[bid]
    assert ch.source_id == 1 and ch.offset == 500000 - 600000
    c = fe.chans[ch.chan_id]
    assert c["src"] == hi["parent_chan"] and c["D"] == 48          # int(1.2e6 / 12500) / 2
    tb.close()
    assert fe.closed


# ------------------------------------------------------------------ frontend_mode == 'pfb': bin routing
class StubPfbFrontend(StubFrontend):
    def __init__(self, samp_rate, center_freq, device=0):
        super().__init__(samp_rate, center_freq, device)
        self.pfb = None

    def pfb_open(self, n_bins, decim, taps):
        self.pfb = dict(n_bins=n_bins, decim=decim, ntaps=len(taps))

    def pfb_tap_open(self, bin_, gr_phase=True):
        assert self.pfb is not None and 0 <= bin_ < self.pfb["n_bins"]
        cid = self.next
        self.next += 1
        self.chans[cid] = dict(cr=None, off=0.0, D=1, bin=bin_, gr_phase=gr_phase)
        return cid


def test_pfb_mode_routes_on_grid_requests_to_bins_and_the_rest_to_the_direct_kernel():
    """The intent of connect_channel_pfb (/root/reference/rc_frontend/receiver.py:343-383): bin = round(offset /
    grid), negative bins wrap, off-grid requests take the other path.  Here the bank is built from the channel
    filter itself (20 Msps, cr 12.5 kHz: 1600 bins, decim 800, 2909 taps), so a bin needs no second stage."""
    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=20000000)},
                                frontend_mode="pfb")
    StubFrontend.instances = []
    tb = receiver.receiver(cfg, frontend_factory=StubPfbFrontend)
    fe = StubFrontend.instances[0]
    assert fe.pfb == dict(n_bins=1600, decim=800, ntaps=2909)
    plan = tb.sources[0]["pfb"]
    assert plan["grid"] == 12500.0 and plan["n_bins"] == 1600
    # on the raster, above and below the centre
    b1, _ = tb.connect_channel(12500, 855000000 + 80 * 12500)
    b2, _ = tb.connect_channel(12500, 855000000 - 3 * 12500)
    assert fe.chans[tb.channels[b1].chan_id]["bin"] == 80 and tb.channels[b1].pfb_bin == 80
    assert fe.chans[tb.channels[b2].chan_id]["bin"] == 1600 - 3            # receiver.py:373-375: wrapped
    assert tb.channels[b1].decim == 800 and tb.channels[b1].ntaps == 2909 and tb.channels[b1].out_rate == 25000.0
    # off the raster (6.25 kHz offset) and another channel rate: the direct xlating FIR
    b3, _ = tb.connect_channel(12500, 855000000 + 6250)
    assert tb.channels[b3].pfb_bin is None and fe.chans[tb.channels[b3].chan_id]["cr"] == 12500
    b4, _ = tb.connect_channel(6250, 855000000 + 12500)
    assert tb.channels[b4].pfb_bin is None and fe.chans[tb.channels[b4].chan_id]["cr"] == 6250
    # an idle bin channel re-used for another frequency moves to that bin (or to the direct kernel)
    tb.release_channel(b1)
    b5, _ = tb.connect_channel(12500, 855000000 + 81 * 12500)
    assert b5 == b1 and tb.channels[b1].pfb_bin == 81 and fe.chans[tb.channels[b1].chan_id]["bin"] == 81
    tb.release_channel(b1)
    b6, _ = tb.connect_channel(12500, 855000000 + 81 * 12500 + 100)
    assert b6 == b1 and tb.channels[b1].pfb_bin is None and fe.chans[tb.channels[b1].chan_id]["off"] == 81 * 12500 + 100
    assert len(fe.chans) == 4                                               # rebuilt channels were closed
    # a 6.25 kHz raster is a config knob
    cfg2 = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=20000000)},
                                 frontend_mode="pfb", pfb_grid=6250)
    tb2 = receiver.receiver(cfg2, frontend_factory=StubPfbFrontend)
    assert StubFrontend.instances[-1].pfb["n_bins"] == 3200
    b7, _ = tb2.connect_channel(12500, 855000000 - 6250)
    assert tb2.channels[b7].pfb_bin == 3199
    # a source rate with no kernel for its bank (2.4 Msps: 192 bins): everything direct, nothing breaks
    cfg3 = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855050000, samp_rate=2400000)},
                                 frontend_mode="pfb")
    tb3 = receiver.receiver(cfg3, frontend_factory=StubPfbFrontend)
    assert tb3.sources[0]["pfb"] is None and StubFrontend.instances[-1].pfb is None
    b8, _ = tb3.connect_channel(12500, 854987500)
    assert tb3.channels[b8].pfb_bin is None


def test_pfb_mode_parity_budget_routes_high_offset_bins_to_the_direct_kernel():
    """the bank's tap phases are exact, GNU Radio's are float32(i * fwT0) -- a per-bin error
    filter (rcf_pfb_tap_leakage) that grows with |offset| (2.4e-4 .. 9.8e-4 rad grid above ~fs/8).
    A request whose predicted discriminator error exceeds pfb_parity_budget takes the direct kernel; the default
    budget is the north-star bar in the cfg2 environment, pfb_parity_budget = None turns the routing off."""
    from rcf import native
    fs = 20000000
    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=fs)},
                                frontend_mode="pfb")
    tb = receiver.receiver(cfg, frontend_factory=StubPfbFrontend)
    fe = StubFrontend.instances[-1]
    plan = tb.sources[0]["pfb"]
    par = plan["parity"]
    assert par["budget"] == 1e-4 and abs(par["gain"] - 25000.0 / (2 * math.pi * 600)) < 1e-9
    served, routed = [], []
    for k in (8, 80, -160, 401, -560, 700, 799, -799):
        bid, _ = tb.connect_channel(12500, 855000000 + k * 12500)
        ch = tb.channels[bid]
        pred = receiver.receiver.pfb_predicted_fm_error(plan, k)
        l2, _ = native.pfb_tap_leakage(fs, 1600, plan["taps"], k % 1600)
        assert abs(pred - par["gain"] * par["margin"] * l2 * 10 ** (par["env_db"] / 20)) < 1e-12
        if pred <= par["budget"]:
            assert ch.pfb_bin == k % 1600 and fe.chans[ch.chan_id]["bin"] == k % 1600
            served.append(k)
        else:
            assert ch.pfb_bin is None and fe.chans[ch.chan_id]["cr"] == 12500 and fe.chans[ch.chan_id]["off"] == k * 12500
            routed.append(k)
    # low offsets stay on the bank, the band edge goes direct
    assert 8 in served and 80 in served and 799 in routed and -799 in routed
    # ... and none of it is silent: the receiver's metrics count how the open channels are served
    tb.connect_channel(12500, 855000000 + 8 * 12500 + 300)            # off the bank's grid
    m = tb.metrics()
    assert m["rcf_pfb_served_by_bank"] == len(served) and m["rcf_pfb_direct_parity_budget"] == len(routed)
    assert m["rcf_pfb_direct_off_grid"] == 1 and "parity budget" in tb.channels[bid].route
    # the prediction is monotone in the float32 exponent ranges: no bin above fs/4 is cheaper than one below fs/32
    assert receiver.receiver.pfb_predicted_fm_error(plan, 799) > 4 * receiver.receiver.pfb_predicted_fm_error(plan, 40)
    # retune of a served channel to a routed bin moves it to the direct kernel, same object
    bid, _ = tb.connect_channel(12500, 855000000 + 16 * 12500)
    ch = tb.channels[bid]
    assert ch.pfb_bin == 16
    ch.set_offset(799 * 12500.0)
    assert ch.pfb_bin is None and fe.chans[ch.chan_id]["off"] == 799 * 12500.0
    ch.set_offset(799 * 12500.0 + 50.0)                       # direct stays direct: retuned in place
    assert fe.chans[ch.chan_id]["off"] == 799 * 12500.0 + 50.0
    # a failed open leaves the object as it was
    old_id = ch.chan_id
    def boom(*a, **k):
        raise RuntimeError("no slot")
    fe.pfb_tap_open, keep = boom, fe.pfb_tap_open
    with pytest.raises(RuntimeError):
        ch.set_offset(16 * 12500.0)
    assert ch.chan_id == old_id and ch.pfb_bin is None and ch.offset == 799 * 12500.0 + 50.0
    fe.pfb_tap_open = keep
    # routing off: every on-grid request is a bin
    cfg2 = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=fs)},
                                 frontend_mode="pfb", pfb_parity_budget=None)
    tb2 = receiver.receiver(cfg2, frontend_factory=StubPfbFrontend)
    bid, _ = tb2.connect_channel(12500, 855000000 + 799 * 12500)
    assert tb2.channels[bid].pfb_bin == 799


def test_tap_leakage_matches_the_float32_phase_model():
    """rcf_pfb_tap_leakage against a numpy restatement of its definition: d[i] = float32(i * float32(2 pi f_k / fs))
    - 2 pi k i / NB, const = h^2-weighted mean, leak = |h (e^{j (d - const)} - 1)|_2; and against the oracle's own
    GNU-Radio composite taps (the error filter IS composite_GR - e^{j const} composite_exact)."""
    from oracle import grspec as G
    from rcf import native
    fs, nb = 20e6, 1600
    D, taps = G.channel_params(fs, 12500)
    i = np.arange(len(taps), dtype=np.float64)
    h = taps.astype(np.float64)
    for k in (0, 1, 80, 401, 799, 800, 801, 1163, 1599):
        ks = k if k < nb // 2 else k - nb
        fw = np.float32(2 * np.pi * (ks * fs / nb) / fs)
        th = (np.arange(len(taps), dtype=np.float32) * fw).astype(np.float64)
        d = th - 2 * np.pi * ks * i / nb
        c = np.sum(h * h * d) / np.sum(h * h)
        want = np.sqrt(np.sum(np.abs(h * (np.exp(1j * (d - c)) - 1)) ** 2))
        l2, cp = native.pfb_tap_leakage(fs, nb, taps, k)
        assert abs(cp - c) < 1e-9 and abs(l2 - want) <= 1e-9 + 1e-6 * want, (k, l2, want, cp, c)
        ct, _ = G.xlating_composite(taps, D, ks * fs / nb, fs)
        g = ct.astype(np.complex128) - np.exp(1j * c) * h * np.exp(2j * np.pi * ks * i / nb)
        assert abs(np.sqrt(np.sum(np.abs(g) ** 2)) - l2) < 3e-8       # float32 taps: 6e-8 relative per tap
    assert native.pfb_tap_leakage(fs, nb, taps, 0) == (0.0, 0.0)
    with pytest.raises(native.RcfError):
        native.pfb_tap_leakage(fs, nb, taps, nb)


def test_receiver_rotator_config_reaches_the_front_end():
    """a receiver's channels iterate GNU Radio's rotator (rcf_set_rotator on every source's front-end, before any
    channel exists) unless config.rotator = 'fast'"""
    class Fe(StubFrontend):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.rotator = None

        def set_rotator(self, exact=True):
            assert not self.chans
            self.rotator = exact

    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=2400000)},
                                frontend_mode="xlat")
    receiver.receiver(cfg, frontend_factory=Fe)
    assert StubFrontend.instances[-1].rotator is True
    cfg.rotator = "exact"
    receiver.receiver(cfg, frontend_factory=Fe)
    assert StubFrontend.instances[-1].rotator is True
    cfg.rotator = "fast"
    receiver.receiver(cfg, frontend_factory=Fe)
    assert StubFrontend.instances[-1].rotator is None
    receiver.receiver(cfg, frontend_factory=StubFrontend)          # a front-end without the knob is left alone


def test_failed_channel_construction_releases_its_egress_port():
    """receiver binds the channel's egress port before building the channel (as the reference's channel
    flowgraph does, channel.py:36); when the build then fails the port must be handed back."""
    cfg = types.SimpleNamespace(sources={0: dict(type="synthetic", center_freq=855000000, samp_rate=2400000)},
                                frontend_mode="xlat")
    tb = receiver.receiver(cfg, frontend_factory=StubFrontend)
    bound, released = [], []
    tb.bind_port = lambda port: bound.append(port) or True
    tb.release_port = released.append
    with pytest.raises(Exception):
        tb.connect_channel(7, 855000000)                    # int(2.4e6 / 7) / 2 is not a decimation: the open fails
    assert len(bound) == 1 and released == bound and not tb.channels
    bid, port = tb.connect_channel(12500, 855012500)        # and the receiver still works
    assert bid in tb.channels and released == bound[:1] and bound[-1] == port


def test_rep_loop_survives_handler_errors_and_egress_isolates_channels():
    """a malformed request must not end serve_zmq (the reference's loop catches and keeps serving,
    receiver.py:686-699); one channel's egress failure must not stop the pump for the others."""
    import sys
    from rcf import egress

    tb = make_receiver()
    srv = protocol.FrontendServer(tb)

    class Again(Exception):
        pass

    class FakeSock:
        def __init__(self, inbox):
            self.inbox, self.sent = list(inbox), []

        def bind(self, addr):
            pass

        def getsockopt(self, opt):
            return b"tcp://0.0.0.0:5555"

        def recv_string(self, flags=0):
            if not self.inbox:
                raise Again()
            return self.inbox.pop(0)

        def send_string(self, s):
            self.sent.append(s)

        def close(self, linger=None):
            self.closed = True

    sock = FakeSock(["connect", "create,x", "offset,1", "scan_mode_set_freq,notanumber", "hb,1", "quit,1"])
    termed = []
    fake_zmq = types.SimpleNamespace(Context=lambda: types.SimpleNamespace(socket=lambda kind: sock, term=lambda: termed.append(1)),
                                     REP=4, LAST_ENDPOINT=32, NOBLOCK=1, Again=Again)
    sys.modules["zmq"] = fake_zmq
    try:
        srv.serve_zmq(stop=lambda: not sock.inbox and len(sock.sent) >= 6)
    finally:
        del sys.modules["zmq"]
    assert len(sock.sent) == 6                                    # every request answered: the REP socket never wedges
    assert sock.closed and termed == [1]                          # ... and the loop, once stopped, gives its socket back
    assert sock.sent[0].startswith("connect,") and sock.sent[-1] == "quit,1"

    # egress: channel A's socket raises on send, channel B keeps flowing
    class Chan:
        def __init__(self, port):
            self.port, self.chan_id, self.reads = port, 1, 0

        def read_iq(self):
            self.reads += 1
            return np.ones(4, dtype=np.complex64)

        def read_fm(self, gain):
            return np.zeros(0, dtype=np.float32)

    class Sock:
        def __init__(self, port):
            self.port, self.sent, self.closed = port, 0, False

        def send(self, payload):
            if self.port == 11111:
                raise OSError("peer gone")
            self.sent += len(payload)

        def close(self):
            self.closed = True

    socks = {}

    def factory(port):
        if port == 12345:
            raise OSError("address already in use")
        socks[port] = Sock(port)
        return socks[port]

    tbx = types.SimpleNamespace(access_lock=tb.access_lock, channels={"a": Chan(11111), "b": Chan(22222)},
                                bind_port=None)
    pump = egress.EgressPump(tbx, socket_factory=factory)
    assert tbx.bind_port is not None and tbx.bind_port(12345) is False and tbx.bind_port(23456) is True
    # a port that was bound but never got a channel (construction failed after bind_port) is released
    # explicitly by the receiver, or swept by the next pump pass -- not held for the life of the process
    assert 23456 in pump.prebound and tbx.release_port is not None
    assert tbx.bind_port(34567) is True
    tbx.release_port(34567)
    assert socks[34567].closed and 34567 not in pump.prebound
    pump.pump_once()
    assert socks[23456].closed and not pump.prebound          # no live channel on 23456: swept
    pump.pump_once()
    assert pump.errors == 2 and socks[22222].sent == 64 and pump.bytes_out == 64


def test_report_offset_survives_the_reply_that_wedges_the_reference():
    """frontend_connector.py:170-185: a reply that is neither 'offset' nor 'na' falls off the reference's if / elif chain
    with thread_lock still held -- its next call never returns (which is why the random golden scripts leave that reply
    out).  The mirror returns the same value (None) and stays usable."""
    sent, replies = [], []
    fc = FC.frontend_connector("parent-uuid", FakeRCM(), transport_factory=lambda h, p: ScriptedSocket(replies, sent),
                               heartbeat=False)
    replies[:] = ["connect,7", "create,u-1,12345"]
    assert fc.create_channel(12500, 855000000) == ("u-1", "12345")
    replies[:] = ["fail,7"]
    assert fc.report_offset(0.25) is None
    replies[:] = ["offset,7"]
    assert fc.report_offset(0.25) is True
    replies[:] = ["release,u-1"]
    assert fc.release_channel() == "u-1"


HANDLER_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "handler.json")))


@pytest.mark.parametrize("sess", HANDLER_GOLD["sessions"], ids=lambda s: "seed%d" % s["seed"])
def test_server_handler_replays_the_references_own_sessions(sess):
    """tests/golden/handler.json was made by RUNNING the reference's nested handler(msg, tb) (rc_frontend/receiver.py:503-
    614, extracted with ast: tests/golden/make_handler_goldens.py) on seeded random sessions against a recording top
    block.  FrontendServer.handle must give the same reply -- or raise where the reference raises --, make the same
    calls on the top block in the same order, and leave the same client tables, message by message."""
    class FakeTB:
        def __init__(self):
            self.calls, self.n, self.channels, self.script = [], 0, {}, []

        def connect_channel(self, channel_rate, freq):
            self.calls.append(["connect_channel", channel_rate, freq])
            if not self.script.pop(0):                           # what the generator's top block did on this very call
                raise Exception("Unable to find source for frequency %s" % freq)
            self.n += 1
            self.channels["blk-%d" % self.n] = object()
            return "blk-%d" % self.n, 10000 + self.n

        def release_channel(self, block_id):
            self.calls.append(["release_channel", block_id])
            return True

        def source_offset(self, block_id, offset):
            self.calls.append(["source_offset", block_id, offset])
            return True

        def scan_mode_set_freq(self, freq):
            self.calls.append(["scan_mode_set_freq", freq])
            return True

    tb = FakeTB()
    srv = protocol.FrontendServer(tb, clock=lambda: 0.0)
    for st in sess["steps"]:
        tb.script[:] = list(st["connect_ok"])
        del tb.calls[:]
        try:
            reply, exc = srv.handle(st["msg"]), None
        except Exception as e:
            reply, exc = None, type(e).__name__
        assert exc == st["raises"], (st["msg"], exc, st["raises"])
        assert reply == st["reply"], (st["msg"], reply, st["reply"])
        assert tb.calls == st["calls"], (st["msg"], tb.calls, st["calls"])
        assert {str(k): list(v) for k, v in srv.clients.items()} == st["clients"], st["msg"]
        assert sorted(str(k) for k in srv.client_hb) == st["client_hb"], st["msg"]
        assert srv.client_num == st["client_num"]


@pytest.mark.parametrize("case", GOLD["publisher"], ids=lambda c: "index_%s" % c["index"])
def test_registry_record_is_the_references_record(case):
    """tests/golden/protocol.json 'publisher': the record rc_frontend/redis_channel_publisher.py itself wrote (run here with
    a recording redis and a ZMQ stub).  The mirror must SADD the same set, SET the same key, and the JSON must carry the
    same keys with the same value types -- and the same values wherever they are not host- or time-dependent."""
    ops = []

    class Rec:
        def sadd(self, k, v): ops.append(["sadd", k, v])
        def set(self, k, v): ops.append(["set", k, v])

    sources = {int(k): v for k, v in case["sources"].items()}
    pub = registry.redis_channel_publisher(sources=sources, channels={i: i for i in range(case["n_channels"])},
                                           port=case["port"], index=case["index"], client=Rec(), start_thread=False)
    pub.publish_once()
    assert [[o[0]] + ([o[1]] if o[0] == "sadd" else []) for o in ops] == [o for o in case["ops"] if o[0] != "execute"]
    rec = json.loads(ops[1][2])
    assert ops[0][2] == ops[1][1] == rec["instance_uuid"]
    volatile = ("instance_uuid", "start_time", "current_time", "hostname", "pid", "address")
    got = {k: (type(v).__name__ if k in volatile else v) for k, v in rec.items()}
    assert got == case["record"]


RECV_GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "receiver.json")))


@pytest.mark.parametrize("sess", RECV_GOLD["sessions"], ids=lambda s: "%s-%d" % (s["config"], s["seed"]))
def test_receiver_replays_the_references_own_sessions(sess):
    """tests/golden/receiver.json was made by RUNNING rc_frontend/receiver.py's receiver class (GNU Radio, UHD and ZeroMQ
    replaced by stand-ins: tests/golden/make_receiver_goldens.py) on seeded random sessions of connect_channel /
    release_channel / source_offset.  rcf.receiver must choose the same source, compute the same offset, re-use the same
    idle channel, draw the same port, accumulate the same drift correction and refuse the same requests, call by call.
    One deliberate difference: with receiver_split2 the reference cannot create a channel at all (its half-band sources
    are copied before 'source_id' is set: KeyError) -- there only the source table and the out-of-range refusals are held
    against it; the working split2 path is tested against the two-stage oracle on the GPU."""
    cfgd = RECV_GOLD["configs"][sess["config"]]
    cfg = types.SimpleNamespace(sources={int(i): dict(s, type="synthetic") for i, s in cfgd["sources"].items()},
                                frontend_mode="xlat", receiver_split2=cfgd["split2"], scan_mode=cfgd.get("scan_mode", False))
    StubFrontend.instances = []
    random.seed(sess["seed"])
    tb = receiver.receiver(cfg, frontend_factory=StubFrontend)
    order = []

    def snapshot():
        chans = []
        for bid in order:
            if bid in tb.channels:
                c = tb.channels[bid]
                chans.append({"n": order.index(bid), "source_id": c.source_id, "channel_rate": c.channel_rate, "offset": c.offset,
                              "in_use": c.in_use, "port": c.port, "idle": c.channel_close_time != 0})
        return {"channels": chans,
                "accumulated": {str(i): tb.sources[i].get("accumulated_offset") for i in tb.sources},
                "sources": {str(i): [tb.sources[i]["center_freq"], tb.sources[i]["samp_rate"]] for i in tb.sources}}

    try:
        for st in sess["steps"]:
            call = st["call"]
            ref_broken = st["raises"] is not None and st["raises"].startswith("KeyError")      # split2 in the reference
            try:
                if call[0] == "connect_channel":
                    bid, port = tb.connect_channel(call[1], call[2])
                    if bid not in order and bid is not False:
                        order.append(bid)
                    ret = [order.index(bid) if bid in order else bid, port]
                elif call[0] == "release_channel":
                    ret = tb.release_channel(order[call[1]] if 0 <= call[1] < len(order) else "no-such-id")
                else:
                    ret = tb.source_offset(order[call[1]] if 0 <= call[1] < len(order) else "no-such-id", call[2])
                exc = None
            except Exception as e:
                ret, exc = None, "%s: %s" % (type(e).__name__, e)
            if cfgd["split2"]:
                assert snapshot()["sources"] == st["after"]["sources"]
                if not ref_broken and call[0] == "connect_channel":
                    assert exc == st["raises"], (call, exc, st["raises"])                      # out of every source's band
                continue
            assert exc == st["raises"], (call, exc, st["raises"])
            assert ret == st["returns"], (call, ret, st["returns"])
            assert snapshot() == st["after"], (call, snapshot(), st["after"])
            if st.get("design"):
                # a new channel: the decimation and the filter design arguments rc_frontend/channel.py handed to GNU
                # Radio (read off the stand-ins) against the C ABI's own rule and design call for that (rate, rate)
                from rcf import native
                d = st["design"]
                ch = tb.channels[order[ret[0]]]
                D, T = native.channel_params(d["samp_rate"], call[1])
                assert D == d["decim"] and d["window_is_hamming"] and d["offset"] == ch.offset
                assert d["low_pass_2"] == [1.0, float(d["samp_rate"]), call[1] / 2, call[1] / 2, 20.0]
                assert T == len(native.design_low_pass_2(*d["low_pass_2"]))
            # what the reference told its SDR to tune to is centre + accumulated (+ the source's static offset): the
            # mirror applies the same Hz to the channels' NCOs instead -- the accumulated value is the comparison
            for i, freqs in st["tuned"].items():
                src = tb.sources[int(i)]
                assert freqs[-1] == src["center_freq"] + src["accumulated_offset"] + src.get("offset", 0)
                assert src["block"].shift == pytest.approx(src["accumulated_offset"])
    finally:
        tb.close()


@pytest.mark.parametrize("case", GOLD["rcm_poll"], ids=lambda c: "idx_%s_n%d" % (c["index"], len(c["members"])))
def test_manager_poll_is_the_references_pass(case):
    """One pass of /root/reference/redis_channelizer_manager.py's manager_loop, run here over seeded random registry contents
    (tests/golden/protocol.json 'rcm_poll').  The mirror removes the same records from Redis (5 s, strictly older) and keeps
    the same live ones under the same index filter.  Two things the reference does are NOT mirrored, both slips in its
    loop: an expired record stays in ITS table until the next pass (`del channelizer[deletion]` deletes from the wrong
    object), and a key whose record is missing takes on the record parsed just before it (`channelizer` is not reset)."""
    table = [m.encode() for m in case["members"]]
    removed = []

    class R:
        def smembers(self, k): return set(table)
        def get(self, k): return case["records"].get(k.decode() if isinstance(k, bytes) else k)
        def srem(self, k, v): removed.append(["srem", k, v])
        def delete(self, k): removed.append(["delete", k])

    mgr = registry.redis_channelizer_manager(index=case["index"], clients=[R()], start_thread=False)
    mgr.poll_once(now=case["now"])
    expired = {r[2] for r in case["removed"] if r[0] == "srem"}
    ghosts = {k for k in case["channelizers"] if k not in case["records"]}      # the reference's carried-over record
    assert sorted(map(tuple, removed)) == sorted(tuple(r) for r in case["removed"] if r[-1] not in ghosts)
    want = {k: v for k, v in case["channelizers"].items() if k not in expired and k not in ghosts}
    assert mgr.channelizers == want


@pytest.mark.parametrize("case", GOLD["heartbeat"], ids=lambda c: "-".join(c["beats"]))
def test_heartbeat_loop_is_the_references(case):
    """frontend_connector.py:197-229 run synchronously here (sleeps counted, not slept) over scripted beats -- answered,
    refused ('fail': the channelizer forgot the client) or unanswered (five receive timeouts) -- then its quit.  The
    mirror's beats send the same requests, reconnect at the same points to the same address and end with the same
    client id."""
    sent, connects = [], []
    queue = []

    class Sock:
        def send_string(self, s): sent.append(s)
        def recv_string(self):
            v = queue.pop(0)
            if v is None:
                raise Exception("timeout")
            return v
        def close(self): pass

    def factory(host, port):
        connects.append("tcp://%s:%s" % (host, port))
        return Sock()

    fc = FC.frontend_connector("parent-uuid", FakeRCM(), transport_factory=factory, heartbeat=False)
    queue[:] = ["connect,7", "create,u-1,12345"]
    assert fc.create_channel(12500, 855000000) == ("u-1", "12345")
    del sent[:], connects[:]
    for v in case["replies"]:
        queue += [None] * 5 if v is None else [v]
    for _ in case["beats"]:
        fc.heartbeat_once()
    fc._call("quit", fc.my_client_id)
    assert sent == case["requests"]
    assert connects == case["connects"]
    assert fc.my_client_id == case["client_id_after"]


@pytest.mark.parametrize("case", HANDLER_GOLD["ticks"], ids=lambda c: "seed%d" % c["seed"])
def test_server_tick_is_one_turn_of_the_references_main_loop(case):
    """tests/golden/handler.json 'ticks': the `while 1:` statement under rc_frontend/receiver.py's __main__ (:620-699) lifted
    out with ast and run for one turn at a scripted time over seeded random client / heartbeat / channel tables.
    FrontendServer.tick + receiver.sweep_idle_channels must release the same channels of the same expired clients, destroy
    the same idle channels, and leave the same tables and timers.  Where the reference's loop dies (a heartbeat entry
    without a client entry: KeyError out of the main loop) the mirror must simply not."""
    import threading
    b = case["before"]
    released, destroyed = [], []

    class Chan:
        def __init__(self, bid, close_time):
            self.block_id, self.channel_close_time = bid, close_time

        def destroy(self):
            destroyed.append(self.block_id)

    tb = types.SimpleNamespace(access_lock=threading.RLock(), channel_idle_timeout=10,
                               last_channel_cleanup=b["last_channel_cleanup"],
                               channels={k: Chan(k, v) for k, v in b["channels"].items()},
                               release_channel=lambda bid: released.append(bid) or True)
    tb.sweep_idle_channels = lambda now=None: receiver.receiver.sweep_idle_channels(tb, now)
    srv = protocol.FrontendServer(tb, clock=lambda: case["now"])
    srv.clients = {int(k): list(v) for k, v in b["clients"].items()}
    srv.client_hb = {int(k): v for k, v in b["client_hb"].items()}
    srv.last_status = b["last_status"]
    srv.tick(case["now"])
    if case["raises"] is not None:
        return                                                   # the reference crashed here; the mirror came through
    a = case["after"]
    assert released == case["released"]
    assert sorted(destroyed) == sorted(case["destroyed"])
    assert sorted(tb.channels) == a["channels"]
    assert {str(k): list(v) for k, v in srv.clients.items()} == a["clients"]
    assert sorted(str(k) for k in srv.client_hb) == a["client_hb"]
    assert srv.last_status == a["last_status"] and tb.last_channel_cleanup == a["last_channel_cleanup"]


def test_feed_all_without_native_front_ends_feeds_one_by_one():
    """receiver.feed_all: the sources of one receiver as one group block (the reference holds all its sources in one top
    block, rc_frontend/receiver.py:67-70); with front-ends that are not librcf handles -- these stubs -- it is feed() /
    feed_raw() per source, the counters the same"""
    rx = make_receiver()
    got = {0: [], 1: []}
    for i, fe in enumerate(StubFrontend.instances):
        fe.push = (lambda i_: (lambda iq: got[i_].append(("cf32", len(iq)))))(i)
        fe.push_raw = (lambda i_: (lambda raw, fmt, scale, offset=0.0: got[i_].append((fmt, len(raw) // 2, scale, offset))))(i)
    rx.feed_all({0: np.zeros(100, np.complex64), 1: np.zeros(50, np.complex64)})
    rx.feed_all({1: np.zeros(40, np.uint8)}, fmt=1, scale=1 / 128.0, offset=127.4)
    assert got[0] == [("cf32", 100)] and got[1] == [("cf32", 50), (1, 20, 1 / 128.0, 127.4)]
    assert rx.metrics()["rcf_samples_in"] == 170
    rx.close()


def test_egress_pump_keeps_the_iq_batch_when_only_the_fm_batch_fails():
    """the batched IQ read advances every channel's reader position; when the batched discriminator read then
    fails, the IQ samples of that pass must not be read a second time (from the advanced positions: they would be lost) --
    only the discriminator stream falls back to channel-by-channel reads"""
    calls = []

    class FE:
        def chan_read_many(self, ids, what, gain=1.0, cap_each=0):
            calls.append(("many", what))
            if what == "fm":
                raise RuntimeError("gather failed")
            return [np.full(3, 10 + i, dtype=np.complex64) for i in range(len(ids))]

    fe = FE()

    class Chan:
        def __init__(self, port, cid):
            self.port, self.chan_id, self.frontend = port, cid, fe

        def read_iq(self):
            calls.append(("iq", self.chan_id))
            return np.zeros(1, dtype=np.complex64)

        def read_fm(self, gain):
            calls.append(("fm", self.chan_id))
            return np.full(2, float(self.chan_id), dtype=np.float32)

    sent = {}

    class Sock:
        def __init__(self, port):
            self.port = port

        def send(self, payload):
            sent.setdefault(self.port, []).append(payload)

        def close(self):
            pass

    import threading
    from rcf import egress
    tbx = types.SimpleNamespace(access_lock=threading.RLock(), channels={"a": Chan(20000, 1), "b": Chan(20002, 2)}, bind_port=None)
    pump = egress.EgressPump(tbx, socket_factory=Sock, fm_gain=5.0)
    pump.pump_once()
    assert ("iq", 1) not in calls and ("iq", 2) not in calls          # the batch's IQ was used, not re-read
    assert ("fm", 1) in calls and ("fm", 2) in calls                  # the discriminator stream fell back per channel
    assert np.frombuffer(sent[20000][0], dtype=np.complex64)[0] == 10 and np.frombuffer(sent[20002][0], dtype=np.complex64)[0] == 11
    assert np.frombuffer(sent[20000 + pump.fm_port_offset][0], dtype=np.float32)[0] == 1.0
