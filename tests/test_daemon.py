"""The channelizer process (rcf.frontend.Daemon == `python -m rcf.frontend -i <index>`, the reference's
rc_frontend/receiver.py:477-700) over real sockets on 127.0.0.1: control wire, data wire and registry of rcf.transport,
the reference-shaped client (rcf.frontend_connector + redis_channelizer_manager) on the other side.  No GPU here: the
front-end is the oracle's channel arithmetic behind the native.Frontend surface, so what is tested is the PLUMBING --
that the bytes a backend pulls off the data wire are the channel's samples, whole and in order."""
import os
import sys
import threading
import time
import types

import numpy as np
import pytest

from oracle import grspec as G
from rcf import frontend, frontend_connector as FC, registry, sources, transport

FS, CR, FC0 = 250000.0, 12500, 855000000


class OracleFrontend:
    """native.Frontend's surface over oracle.grspec (tests only); incremental: a read computes the new outputs only,
    carrying GNU Radio's rotator state, so it equals xlating_fir_ccc over the whole stream from the channel's start"""

    def __init__(self, samp_rate, center_freq, device=0):
        self.fs, self.n, self.chans, self.next, self.lock = samp_rate, 0, {}, 1, threading.Lock()
        self.x = np.zeros(int(samp_rate * 120), np.complex64)

    def set_rotator(self, exact=True):
        pass

    @property
    def samples_in(self):
        return self.n

    def push(self, iq):
        with self.lock:
            self.x[self.n:self.n + len(iq)] = iq
            self.n += len(iq)

    def chan_open(self, cr, off):
        D, taps = G.channel_params(self.fs, cr)
        ctaps, incr = G.xlating_composite(taps, D, off, self.fs)
        with self.lock:
            cid, self.next = self.next, self.next + 1
            self.chans[cid] = dict(D=D, T=len(taps), ctaps=ctaps, incr=incr, off=off, start=self.n, read=0,
                                   phase=np.complex64(1.0), count=0)
        return cid

    def chan_info(self, cid):
        c = self.chans[cid]
        return dict(decim=c["D"], ntaps=c["T"], out_rate=self.fs / c["D"], offset_hz=c["off"])

    def chan_start(self, cid):
        return self.chans[cid]["start"]

    def chan_set_offset(self, cid, off):
        self.chans[cid]["off"] = off

    def chan_close(self, cid):
        with self.lock:
            del self.chans[cid]

    def chan_read_iq(self, cid, max_samples=1 << 20):
        with self.lock:
            c = self.chans[cid]
            avail = self.n - c["start"]
            n_out = 0 if avail <= 0 else (avail - 1) // c["D"] + 1
            k0, k1 = c["read"], n_out
            if k1 <= k0:
                return np.zeros(0, np.complex64)
            T, D = c["T"], c["D"]
            lo = c["start"] + k0 * D - (T - 1)                        # first sample output k0 touches
            seg = self.x[max(lo, c["start"]): c["start"] + (k1 - 1) * D + 1].astype(np.complex128)
            seg = np.concatenate([np.zeros(max(lo, c["start"]) - lo, np.complex128), seg])   # zero history before the start
            idx = (np.arange(k1 - k0) * D)[:, None] + np.arange(T)[None, :]
            v = (seg[idx] @ c["ctaps"][::-1].astype(np.complex128)).astype(np.complex64)
            ph, c["phase"], c["count"] = G.rotator_phases(c["incr"], len(v), c["phase"], c["count"])
            f32 = np.float32
            y = np.empty(len(v), np.complex64)
            y.real = v.real.astype(f32) * ph.real.astype(f32) - v.imag.astype(f32) * ph.imag.astype(f32)
            y.imag = v.real.astype(f32) * ph.imag.astype(f32) + v.imag.astype(f32) * ph.real.astype(f32)
            c["read"] = k1
        return y

    def source_shift(self, d):
        pass

    def close(self):
        pass


def _config(block_ms=20.0, **src):
    s = dict(type="synthetic", center_freq=FC0, samp_rate=int(FS), seed=7, tile_samples=1 << 16, block_ms=block_ms,
             carriers=[dict(f_off=25000.0, f_mod=700.0, dev=2500.0, snr_db=30.0)])
    s.update(src)
    return types.SimpleNamespace(sources={0: s}, frontend_mode="xlat", receiver_split2=False)


@pytest.fixture
def daemon(tmp_path):
    d = frontend.Daemon(_config(), index=0, transport="tcp", registry="dir:%s" % (tmp_path / "reg"), bind="127.0.0.1",
                        frontend_factory=OracleFrontend)
    t = threading.Thread(target=d.serve_forever, daemon=True)
    t.start()
    yield d, transport.DirRegistryClient(str(tmp_path / "reg"))
    d.stop()
    t.join(timeout=10)


def _manager(client, timeout=10):
    mgr = registry.redis_channelizer_manager(clients=[client], start_thread=False)
    t0 = time.time()
    while not mgr.channelizers:
        assert time.time() - t0 < timeout, "the daemon never appeared in the registry"
        time.sleep(0.1)
        mgr.poll_once()
    return mgr


def _align(got, want, probe=48):
    """index k0 of `want` at which the received samples start (the data wire has no timestamps: a subscriber joins
    wherever the stream is -- channel.py:36's PUB socket gives no more)"""
    g = got[:probe]
    win = np.lib.stride_tricks.sliding_window_view(want[: len(want) - len(got) + probe], probe)
    return int(np.argmin(np.abs(win - g).sum(axis=1)))


def test_create_channel_and_pull_the_oracles_samples_off_the_data_wire(daemon):
    d, reg = daemon
    mgr = _manager(reg)
    rec = next(iter(mgr.channelizers.values()))
    assert rec["port"] == d.port and rec["index"] == 0 and rec["sources"] == [[FC0, int(FS)]]
    fc = FC.frontend_connector("test", mgr, transport_factory=transport.tcp_req_factory)
    try:
        chan, port = fc.create_channel(CR, FC0 + 25000)
        assert chan and isinstance(port, str)                     # the reference hands the port back as a string
        sub = transport.TcpSubSocket(fc.host, port)
        n = 6000
        got = np.frombuffer(sub.recv_exact(8 * n), dtype=np.complex64)
        sub.close()
        t0 = time.time()                                          # the record is refreshed once a second
        while chan not in next(iter(mgr.channelizers.values())).get("rcf_channel_starts", {}):
            assert time.time() - t0 < 5
            time.sleep(0.1)
            mgr.poll_once()
        start, decim = next(iter(mgr.channelizers.values()))["rcf_channel_starts"][chan]
        assert decim == 10
        # the stream the daemon's paced source delivers: the config's tile, looped
        tile = sources.synthetic_tile(d.tb.realsources[0])
        need = start + (n + 40000) * decim
        x = np.tile(tile, need // len(tile) + 2)[start:need]
        want = G.xlating_fir_ccc(x, decim, G.channel_params(FS, CR)[1], 25000.0, FS)
        k0 = _align(got, want)
        ref = want[k0:k0 + n]
        err = np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2))
        assert err < 1e-6, (k0, err)
        assert np.mean(np.abs(ref) ** 2) > 0.3                    # the carrier is in there: not a match of noise to noise
        assert fc.release_channel() == chan
    finally:
        fc.exit()
    src = d.sources[0].stats()
    assert src["blocks"] > 5 and src["error"] is None


def test_silent_client_loses_its_channel_after_five_seconds_and_the_record_expires(daemon):
    d, reg = daemon
    mgr = _manager(reg)
    req = transport.TcpReqSocket("127.0.0.1", d.port)
    req.send_string("connect")
    cid = req.recv_string().split(",")[1]
    req.send_string("create,%s,%d,%d" % (cid, CR, FC0 - 12500))
    verb, block_id, port = req.recv_string().split(",")
    assert verb == "create" and d.tb.channels[block_id].in_use
    req.send_string("hb,%s" % cid)
    assert req.recv_string() == "hb,%s" % cid
    t0 = time.time()                                              # ... and now the client goes silent (receiver.py:654-668)
    while d.tb.channels[block_id].in_use:
        assert time.time() - t0 < 9, "heartbeat expiry did not release the channel"
        time.sleep(0.1)
    assert time.time() - t0 > 4.5
    req.send_string("hb,%s" % cid)
    assert req.recv_string() == "fail,%s" % cid                  # the server forgot the client
    req.close()
    # the idle sweep (10 s idle, checked with the 10 s status: receiver.py:622-648) destroys it; speed the clocks up
    d.tb.channel_idle_timeout = 0.2
    time.sleep(0.4)                                               # idle for longer than that
    d.tb.last_channel_cleanup -= 100
    d.server.last_status -= 100
    t0 = time.time()
    while block_id in d.tb.channels:
        assert time.time() - t0 < 5, "idle sweep did not destroy the released channel"
        time.sleep(0.05)
    # the registry record: refreshed every second while the daemon lives, gone 5 s after it stops (manager.py:106-110)
    mgr.poll_once()
    assert len(mgr.channelizers) == 1
    d.publisher.continue_running = False
    time.sleep(1.2)
    mgr.poll_once(now=time.time() + 6)
    assert mgr.channelizers == {} and reg.smembers("channelizers") == set()


def test_paced_source_keeps_wall_clock_rate_and_wire_formats_round_trip():
    fed = []
    tb = types.SimpleNamespace(feed=lambda sid, blk: fed.append(("cf32", len(blk))),
                               feed_raw=lambda sid, raw, fmt, scale, off: fed.append((fmt, len(raw) // 2, raw[:8].copy(), scale, off)))
    now = [0.0]
    slept = []

    def sleep(dt):
        slept.append(dt)
        now[0] += dt
    src = dict(type="synthetic", samp_rate=250000, seed=3, tile_samples=10000, wire="u8", block_ms=10.0)
    p = sources.PacedSource(tb, 0, src, pinned=False, clock=lambda: now[0], sleep=sleep)
    p.run(max_blocks=12)
    assert p.blocks == 12 and p.samples == 12 * 2500 and p.late == 0
    assert abs(now[0] - 12 * 0.01) < 1e-9                         # block k is delivered at t0 + (k + 1) * 10 ms -- when its last sample exists --, never earlier
    fmt, n, head, scale, off = fed[0]
    assert fmt == 1 and n == 2500
    tile = sources.synthetic_tile(src)
    back = (head.astype(np.float32) - off) * scale               # what rcf_push_raw computes on the GPU
    assert np.max(np.abs(back - tile[:4].view(np.float32))) <= scale / 2 + 1e-6
    # a slow consumer: every delivery takes 25 ms of a 10 ms block -> late blocks counted, nothing dropped
    tb.feed_raw = lambda *a: now.__setitem__(0, now[0] + 0.025)
    q = sources.PacedSource(tb, 0, src, pinned=False, clock=lambda: now[0], sleep=sleep)
    q.run(max_blocks=10)
    assert q.blocks == 10 and q.samples == 25000 and q.late >= 6 and q.max_lag_s > 0.1


def test_file_source_replays_a_capture(tmp_path):
    x = (np.arange(1000) + 1j * np.arange(1000)).astype(np.complex64)
    path = tmp_path / "cap.cf32"
    x.tofile(path)
    got = []
    tb = types.SimpleNamespace(feed=lambda sid, blk: got.append(blk.copy()))
    p = sources.PacedSource(tb, 0, dict(type="file", path=str(path), format="cf32", samp_rate=1e6, loop=False, block_ms=0.3),
                            pinned=False, clock=lambda: 0.0, sleep=lambda dt: None)
    p.run()
    assert np.array_equal(np.concatenate(got), x) and [len(g) for g in got] == [300, 300, 300, 100]
    with pytest.raises(ValueError):
        sources.PacedSource(tb, 0, dict(type="rtlsdr", samp_rate=1e6), pinned=False)


@pytest.mark.skipif(not frontend.have("zmq"), reason="pyzmq not installed")
def test_daemon_over_real_zeromq(tmp_path):
    """the same conversation through real ZeroMQ REQ/REP + PUB/SUB, wherever pyzmq exists"""
    import zmq
    d = frontend.Daemon(_config(), index=0, transport="zmq", registry="dir:%s" % (tmp_path / "reg"), bind="127.0.0.1",
                        frontend_factory=OracleFrontend)
    t = threading.Thread(target=d.serve_forever, daemon=True)
    t.start()
    try:
        mgr = _manager(transport.DirRegistryClient(str(tmp_path / "reg")))
        fc = FC.frontend_connector("test", mgr)                   # the default link IS pyzmq REQ
        chan, port = fc.create_channel(CR, FC0 + 25000)
        assert chan
        sub = zmq.Context.instance().socket(zmq.SUB)
        sub.setsockopt(zmq.SUBSCRIBE, b"")
        sub.setsockopt(zmq.RCVTIMEO, 5000)
        sub.connect("tcp://127.0.0.1:%s" % port)
        buf = b""
        while len(buf) < 8 * 3000:
            buf += sub.recv()
        assert len(buf) % 8 == 0
        fc.release_channel()
        fc.exit()
    finally:
        d.stop()
        t.join(timeout=10)


@pytest.mark.skipif(not frontend.have("redis"), reason="redis-py not installed")
def test_registry_against_a_real_redis_server():
    import redis
    client = redis.StrictRedis(host="127.0.0.1", port=6379, db=0)
    try:
        client.ping()
    except Exception:
        pytest.skip("no redis server on 127.0.0.1:6379")
    pub = registry.redis_channel_publisher(sources={0: dict(center_freq=FC0, samp_rate=int(FS))}, channels={}, port=12345,
                                           index=3, client=client, start_thread=False)
    pub.publish_once()
    mgr = registry.redis_channelizer_manager(index=3, clients=[client], start_thread=False)
    mgr.poll_once()
    assert mgr.channelizers[pub.instance_uuid]["port"] == 12345
    mgr.poll_once(now=time.time() + 6)
    assert pub.instance_uuid not in mgr.channelizers


def test_the_references_own_smoke_and_speed_test_of_the_api_runs_again(daemon, capsys):
    """frontend_connector.py:232-251 (create + release, then 100 x timed) cannot run in the reference any more; its working
    equivalent ships as `python -m rcf.frontend_connector` and runs here against the daemon over real sockets"""
    d, reg = daemon
    _manager(reg)
    assert FC.main(["--registry", "dir:%s" % reg.root, "--transport", "tcp", "--freq", str(FC0 + 12500), "--rate", str(CR),
                    "-n", "20"]) == 0
    out = capsys.readouterr().out
    assert "function test pass" in out and "speed test" in out


def test_control_wire_survives_garbage_and_serves_many_clients_at_once(daemon):
    """a client that sends an impossible frame length, or half a frame and hangs up, costs itself its connection and
    nobody else anything; requests of several connections interleave (receiver.py:686-699 answers one request per
    turn of its loop, whoever sent it)"""
    import socket
    import struct
    d, reg = daemon
    good = [transport.TcpReqSocket("127.0.0.1", d.port) for _ in range(4)]
    bad = socket.create_connection(("127.0.0.1", d.port))
    bad.sendall(struct.pack("<I", 0x7fffffff) + b"x" * 64)        # 2 GiB frame announced
    half = socket.create_connection(("127.0.0.1", d.port))
    half.sendall(struct.pack("<I", 100) + b"conn")                 # then silence
    ids = []
    for s in good:
        s.send_string("connect")
    for s in good:
        verb, cid = s.recv_string().split(",")
        assert verb == "connect"
        ids.append(cid)
    assert len(set(ids)) == 4
    half.close()
    for s, cid in zip(good, ids):
        s.send_string("hb,%s" % cid)
        assert s.recv_string() == "hb,%s" % cid
    good[0].send_string("create,x")                                 # malformed: the reference logs and keeps serving
    assert good[0].recv_string() == "na"
    good[1].send_string("nonsense")                                 # unknown verb: empty reply, as the reference's handler returns None
    assert good[1].recv_string() == ""
    bad.settimeout(2.0)
    assert bad.recv(16) == b""                                      # the daemon hung up on the garbage sender
    for s, cid in zip(good, ids):
        s.send_string("quit,%s" % cid)
        assert s.recv_string() == "quit,%s" % cid
        s.close()
    bad.close()


def test_python2_indented_site_configs_load(tmp_path):
    """20 of the reference's 22 configs/*.py mix tabs and spaces the way Python 2 allowed (tab = next multiple of eight)
    and raise TabError under Python 3 -- the reference itself cannot import them any more; the channelizer process
    reads them the Python-2 way"""
    p = tmp_path / "config.py"
    p.write_text("class rc_config:\n\tdef __init__(self):\n                self.receiver_split2 = False\n"
                 "\t\tself.sources = {\n\t\t\t0:{\n                                'type': 'rtlsdr',\n"
                 "\t\t\t\t'center_freq': 855500000,\n                                'samp_rate': 10666666,\n\t\t\t}\n\t\t}\n")
    with pytest.raises(TabError):
        compile(p.read_text(), str(p), "exec")
    c = frontend.load_config(str(p))
    assert c.receiver_split2 is False and c.sources[0]["samp_rate"] == 10666666


def test_every_site_config_of_the_reference_can_open_its_channels():
    """tests/golden/reference_configs.json (what configs/*.py ask for, read in the build container): every source of
    every site configuration opens 12.5 kHz channels -- with channel.py:31's rule as Python 3 reads it, or, for the two
    10 666 666 sps deployments, as Python 2 read it (py2_decim / RCF_DECIM_FLOOR)"""
    import json
    from rcf import native
    cfgs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_configs.json")))
    assert len(cfgs) == 22 and sum(not c["imports_under_python3"] for c in cfgs.values()) == 20
    need_floor = set()
    for name, c in cfgs.items():
        for s in c["sources"]:
            if s["samp_rate"] is None:
                continue                                     # (one file leaves the rate to its device discovery)
            rate = s["samp_rate"] / (2 if c["receiver_split2"] else 1)
            try:
                D, T = native.channel_params(rate, 12500)
            except native.RcfError as e:
                assert e.code == native.RCF_ERANGE
                D, T = native.channel_params(rate, 12500, native.DECIM_FLOOR)
                need_floor.add(name)
            assert D == int(rate / 12500) // 2 and T % 2 == 1
    assert need_floor == {"config_denver_massive_p25.py", "config_denver_usrp.py"}


def test_data_wire_drops_whole_messages_for_a_slow_subscriber_and_stays_item_aligned():
    """channel.py:36's PUB socket is lossy at its high-water mark and never blocks the flowgraph; the TCP stand-in behaves
    the same way: a subscriber that stops reading loses WHOLE sends (never part of one: the cf32 stream stays item-aligned
    and every surviving send arrives intact and in order), a fast subscriber next to it loses nothing, and the publisher
    never blocks"""
    import socket
    pub = transport.TcpPubSocket(0, host="127.0.0.1", sndbuf=1 << 16)
    fast = transport.TcpSubSocket("127.0.0.1", pub.port)
    slow = transport.TcpSubSocket("127.0.0.1", pub.port)
    n_msg, items = 400, 2048
    msgs = [np.full(items, k, dtype=np.complex64).tobytes() for k in range(n_msg)]
    got_fast = bytearray()
    fast.sock.settimeout(0.2)
    t0 = time.time()
    for m in msgs:
        pub.send(m)                                           # the slow subscriber's buffers fill up after a few dozen
        try:
            while len(got_fast) < (msgs.index(m) + 1) * len(m):
                got_fast += fast.sock.recv(1 << 20)
        except socket.timeout:
            pass
    assert time.time() - t0 < 20                              # never blocked on the stuck subscriber
    assert bytes(got_fast) == b"".join(msgs)                  # the reader that keeps up has everything
    assert pub.dropped > 0
    slow.sock.settimeout(0.3)
    got = bytearray()
    try:
        while True:
            pub.send(b"")                                     # lets a pending tail out
            chunk = slow.sock.recv(1 << 20)
            if not chunk:
                break
            got += chunk
    except socket.timeout:
        pass
    assert len(got) % (items * 8) == 0 and len(got) < n_msg * items * 8
    seq = np.frombuffer(bytes(got), dtype=np.complex64).reshape(-1, items)
    assert np.all(seq == seq[:, :1])                          # every surviving message intact ...
    ks = seq[:, 0].real.astype(int)
    assert np.all(np.diff(ks) > 0)                            # ... and in order, with gaps where sends were dropped
    for s in (fast, slow):
        s.close()
    pub.close()


def test_a_client_that_never_reads_its_replies_does_not_stall_the_control_loop():
    """replies are written without ever blocking (queued per connection, flushed from the selector's write
    set); a client that stops reading is dropped at a megabyte of backlog, everybody else is served meanwhile"""
    import socket as so
    import struct
    import time as _t
    from rcf import transport
    rep = transport.TcpRepServer("127.0.0.1", 0)
    big = "x" * 8192
    try:
        deaf = so.create_connection(("127.0.0.1", rep.port))
        deaf.setsockopt(so.SOL_SOCKET, so.SO_RCVBUF, 4096)
        frame = struct.pack("<I", 2) + b"hb"
        deaf.sendall(frame * 2000)                           # 2000 requests, 16 MB of replies nobody will read
        good = transport.TcpReqSocket("127.0.0.1", rep.port)
        good.send_string("ping")
        t0 = _t.monotonic()
        answered = None
        while _t.monotonic() - t0 < 5.0 and answered is None:
            rep.poll(lambda m: big if m == "hb" else "pong", 0.001)
            good.sock.settimeout(0.001)
            try:
                answered = good.recv_string()
            except (so.timeout, BlockingIOError, OSError):
                pass
        assert answered == "pong" and _t.monotonic() - t0 < 2.0
        for _ in range(200):
            rep.poll(lambda m: big if m == "hb" else "pong", 0.001)
        assert len(rep.conns) == 1                           # the deaf client is gone, the good one is still there
        good.close()
        deaf.close()
    finally:
        rep.close()
