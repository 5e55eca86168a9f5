"""End to end on the GPU through the reference's own API surface: a backend-style client asks
`frontend_connector.create_channel(rate, freq)`, the front-end (receiver mirror on librcf) is fed
wideband IQ, and the client-side stream equals what the reference's per-channel GNU Radio path would
produce (oracle).  BASELINE configs[0] shape."""
import types

import numpy as np
import pytest

from oracle import cbind as OC
from oracle import grspec as G
from rcf import frontend_connector as FC
from rcf import protocol, receiver, synth

pytestmark = pytest.mark.gpu


class OneChannelizer:
    def get_channelizer_for_frequency(self, f):
        return ("127.0.0.1", 0)


def test_create_channel_feed_read_release(gpu_required):
    x, meta = synth.cfg1(seconds=0.2)
    cfg = types.SimpleNamespace(
        sources={0: dict(type="synthetic", center_freq=meta["center_freq"], samp_rate=int(meta["fs"]))},
        frontend_mode="xlat")
    tb = receiver.receiver(cfg)
    try:
        srv = protocol.FrontendServer(tb)
        fc = FC.frontend_connector("backend-uuid", OneChannelizer(), heartbeat=False,
                                   transport_factory=lambda h, p: protocol.LoopbackTransport(srv))
        cid, port = fc.create_channel(12500, meta["freq"])
        assert cid and isinstance(port, str)
        ch = tb.channels[cid]
        assert ch.decim == 96 and ch.ntaps == 349 and ch.offset == meta["offset"]
        half = len(x) // 2 + 1234
        tb.feed(0, x[:half])
        tb.feed(0, x[half:])
        y = ch.read_iq()
        fm = ch.read_fm(G.p25_fm_gain(25000.0))
        D, taps = G.channel_params(meta["fs"], 12500)
        ct, incr = OC.xlating_composite(taps, D, meta["offset"], meta["fs"])
        yo, fo = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[G.p25_fm_gain(25000.0)])
        assert len(y) == yo.shape[1]
        assert np.sqrt(np.mean(np.abs(y - yo[0]) ** 2) / np.mean(np.abs(yo[0]) ** 2)) < 1e-5
        assert np.sqrt(np.mean((fm - fo[0]) ** 2)) < 1e-4
        # drift report > dead band retunes every channel's NCO (receiver.source_offset)
        assert fc.report_offset(2.0) is True
        assert tb.sources[0]["accumulated_offset"] == 100.0
        tb.feed(0, x[:96 * 50])
        y2 = ch.read_iq()
        assert len(y2) == 50
        assert fc.release_channel() == cid and not ch.in_use
        # the idle channel is re-used for the next request of the same rate (receiver.py:311-319)
        cid2, _ = fc.create_channel(12500, meta["freq"] + 25000)
        assert cid2 == cid and ch.offset == meta["offset"] + 25000
        # 10 s idle sweep destroys released channels
        fc.release_channel()
        assert tb.sweep_idle_channels(now=ch.channel_close_time + 11) == [cid]
        assert tb.channels == {}
    finally:
        tb.close()


def test_egress_pump_publishes_bare_cf32_bytes(gpu_required):
    """Surface (2) of SURVEY 8(b): per-channel PUB of raw cf32 items (rc_frontend/channel.py:36)."""
    from rcf import egress
    x, meta = synth.cfg1(seconds=0.05, seed=31)
    cfg = types.SimpleNamespace(
        sources={0: dict(type="synthetic", center_freq=meta["center_freq"], samp_rate=int(meta["fs"]))},
        frontend_mode="xlat")
    tb = receiver.receiver(cfg)

    class FakePub:
        made = {}

        def __init__(self, port):
            self.port, self.chunks = port, []
            FakePub.made[port] = self

        def send(self, b):
            self.chunks.append(b)

        def close(self):
            self.closed = True
    try:
        bid, port = tb.connect_channel(12500, meta["freq"])
        pump = egress.EgressPump(tb, socket_factory=FakePub, fm_gain=5.0)
        tb.feed(0, x[: len(x) // 2])
        pump.pump_once()
        tb.feed(0, x[len(x) // 2:])
        pump.pump_once()
        iq = np.frombuffer(b"".join(FakePub.made[port].chunks), dtype=np.complex64)
        fm = np.frombuffer(b"".join(FakePub.made[port + 1].chunks), dtype=np.float32)
        D, taps = G.channel_params(meta["fs"], 12500)
        ct, incr = OC.xlating_composite(taps, D, meta["offset"], meta["fs"])
        yo, fo = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[5.0])
        assert len(iq) == yo.shape[1] and len(fm) == len(iq)
        assert np.sqrt(np.mean(np.abs(iq - yo[0]) ** 2) / np.mean(np.abs(yo[0]) ** 2)) < 1e-5
        assert np.sqrt(np.mean((fm - fo[0]) ** 2)) < 1e-4
        tb.release_channel(bid)
        tb.sweep_idle_channels(now=tb.channels[bid].channel_close_time + 11)
        pump.pump_once()
        assert FakePub.made[port].closed and pump.socks == {}
    finally:
        tb.close()


def test_receiver_split2_chain_equals_two_stage_oracle(gpu_required):
    """receiver_split2 (receiver.py:205-237): source -> /2 half-band xlating FIR at -/+fs/4 -> the channel's
    own xlating FIR at the half rate.  The chained device path equals the two GNU Radio blocks in series."""
    x, meta = synth.cfg1(seconds=0.1)
    fs = meta["fs"]
    cfg = types.SimpleNamespace(
        sources={0: dict(type="synthetic", center_freq=meta["center_freq"], samp_rate=int(fs))},
        frontend_mode="xlat", receiver_split2=True)
    tb = receiver.receiver(cfg)
    try:
        bid, _ = tb.connect_channel(12500, meta["freq"])
        ch = tb.channels[bid]
        src = tb.sources[ch.source_id]
        sign = -1.0 if src["center_freq"] < meta["center_freq"] else 1.0
        assert src["samp_rate"] == fs / 2 and ch.decim == 48
        cut = len(x) // 3 + 77
        tb.feed(ch.source_id, x[:cut])
        tb.feed(ch.source_id, x[cut:])
        y = ch.read_iq()
    finally:
        tb.close()
    t1 = G.low_pass_2(1.0, fs, fs / 4, fs / 8, 53.0)
    assert len(t1) == 19
    ct1, incr1 = OC.xlating_composite(t1, 2, sign * fs / 4, fs)
    v1 = G.fir_decim_cc(x, ct1, 2)
    ph1, _, _ = G.rotator_phases(incr1, len(v1))
    h = (v1 * ph1).astype(np.complex64)
    D2, t2 = G.channel_params(fs / 2, 12500)
    ct2, incr2 = OC.xlating_composite(t2, D2, ch.offset, fs / 2)
    v2 = G.fir_decim_cc(h, ct2, D2)
    ph2, _, _ = G.rotator_phases(incr2, len(v2))
    yo = (v2 * ph2).astype(np.complex64)
    assert len(y) == len(yo) > 100
    err = np.sqrt(np.mean(np.abs(y - yo) ** 2)) / np.sqrt(np.mean(np.abs(yo) ** 2))
    assert err < 2e-5
