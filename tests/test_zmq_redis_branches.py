"""Every line of the ZeroMQ / Redis branches of the channelizer in the CPU suite:
rcf.protocol.FrontendServer.serve_zmq (rc_frontend/receiver.py:44-46,686-699), rcf.egress.zmq_pub_factory
(rc_frontend/channel.py:36), rcf.registry's default redis.StrictRedis clients (rc_frontend/redis_channel_publisher.py:27,31;
redis_channelizer_manager.py), rcf.frontend_connector's default REQ link (frontend_connector.py:45-52) and
rcf.frontend.Daemon(transport='zmq', registry='redis') -- over tests/fake_zmq.py and tests/fake_redis.py, API-faithful
in-process stand-ins injected through sys.modules (pyzmq / redis-py are not installable in this image).  The front-end
is the oracle's channel arithmetic (tests/test_daemon.py: OracleFrontend): what is tested is the plumbing."""
import json
import os
import sys
import threading
import time

import numpy as np
import pytest

import fake_redis
import fake_zmq
from oracle import grspec as G
from rcf import frontend, frontend_connector as FC, protocol, registry, sources
from test_daemon import CR, FC0, FS, OracleFrontend, _align, _config


@pytest.fixture
def fakes(monkeypatch):
    fake_zmq.reset()
    fake_redis.reset()
    fake_zmq.Context._instance = None
    monkeypatch.setitem(sys.modules, "zmq", fake_zmq)
    monkeypatch.setitem(sys.modules, "redis", fake_redis)
    yield fake_zmq, fake_redis
    fake_zmq.reset()
    fake_redis.reset()


def test_fake_zmq_has_the_semantics_the_code_is_written_against(fakes):
    zmq, _ = fakes
    ctx = zmq.Context()
    rep = ctx.socket(zmq.REP)
    rep.bind("tcp://0.0.0.0:0")
    ep = rep.getsockopt(zmq.LAST_ENDPOINT).decode()
    assert ep.startswith("tcp://0.0.0.0:") and int(ep.rsplit(":", 1)[1]) >= 40000
    with pytest.raises(zmq.Again):
        rep.recv_string(flags=zmq.NOBLOCK)                        # the reference's main loop polls like this
    req = ctx.socket(zmq.REQ)
    req.setsockopt(zmq.RCVTIMEO, 50)
    req.connect("tcp://127.0.0.1:%s" % ep.rsplit(":", 1)[1])
    req.send_string("connect")
    with pytest.raises(zmq.ZMQError):
        req.send_string("again")                                  # REQ is lock step
    assert rep.recv_string(flags=zmq.NOBLOCK) == "connect"
    with pytest.raises(zmq.ZMQError):
        rep.recv_string(flags=zmq.NOBLOCK)                        # ... and so is REP
    rep.send_string("connect,1")
    assert req.recv_string() == "connect,1"
    nobody = ctx.socket(zmq.REQ)
    nobody.setsockopt(zmq.RCVTIMEO, 30)
    nobody.connect("tcp://127.0.0.1:1")
    nobody.send_string("hello")                                   # queued: zmq does not fail a send to an absent peer
    with pytest.raises(zmq.Again):
        nobody.recv_string()
    pub = ctx.socket(zmq.PUB)
    pub.bind("tcp://0.0.0.0:50001")
    with pytest.raises(zmq.ZMQError):
        ctx.socket(zmq.PUB).bind("tcp://0.0.0.0:50001")           # the receiver's port-retry loop depends on this
    pub.send(b"lost: nobody is subscribed yet")
    sub = ctx.socket(zmq.SUB)
    sub.setsockopt(zmq.SUBSCRIBE, b"")
    sub.setsockopt(zmq.RCVHWM, 3)
    sub.setsockopt(zmq.RCVTIMEO, 20)
    sub.connect("tcp://127.0.0.1:50001")
    for i in range(5):
        pub.send(b"m%d" % i)                                      # no back-pressure: 3 kept, 2 dropped at the HWM
    assert [sub.recv() for _ in range(3)] == [b"m0", b"m1", b"m2"]
    with pytest.raises(zmq.Again):
        sub.recv()
    ctx.term()


def test_default_redis_clients_publish_and_read_the_reference_record(fakes):
    """redis_channel_publisher / redis_channelizer_manager constructed WITHOUT a client: their own
    redis.StrictRedis(host='127.0.0.1', port=6379, db=0) (redis_channel_publisher.py:27,31)"""
    _, redis = fakes
    pub = registry.redis_channel_publisher(sources={0: dict(center_freq=FC0, samp_rate=int(FS))}, channels={"a": 1},
                                           port=4242, index=3, start_thread=False)
    mgr = registry.redis_channelizer_manager(index=3, start_thread=False)
    assert pub.client.connection_args == ("127.0.0.1", 6379, 0) and mgr.clients[0].connection_args == ("127.0.0.1", 6379, 0)
    d = pub.publish_once()
    assert [c[0] for c in redis.calls] == ["SADD", "SET"] and redis.calls[0][1] == "channelizers"
    raw = fake_redis.StrictRedis("127.0.0.1", 6379, 0).get(pub.instance_uuid)
    assert isinstance(raw, bytes) and json.loads(raw)["port"] == 4242 and json.loads(raw)["index"] == 3
    mgr.poll_once()                                               # members and values arrive as bytes, as from redis-py
    assert mgr.channelizers[pub.instance_uuid]["channel_count"] == d["channel_count"] == 1
    assert mgr.get_channelizer_for_frequency(FC0 + 1000)[1] == 4242
    mgr.poll_once(now=time.time() + 6)                            # 5 s expiry: SREM + DELETE
    assert mgr.channelizers == {} and [c[0] for c in redis.calls[-2:]] == ["SREM", "DELETE"]
    assert fake_redis.StrictRedis("127.0.0.1", 6379, 0).smembers("channelizers") == set()


def test_the_channelizer_process_over_zeromq_and_redis(fakes):
    """Daemon(transport='zmq', registry='redis'): create -> bytes off the PUB socket == the oracle's channel -> the
    client goes silent -> heartbeat expiry releases the channel -> quit; the client is rcf.frontend_connector with its
    DEFAULT link (zmq REQ, 1 s timeouts) and the manager with its default redis client"""
    zmq, redis = fakes
    d = frontend.Daemon(_config(), index=0, transport="zmq", registry="redis", bind="127.0.0.1",
                        frontend_factory=OracleFrontend)
    t = threading.Thread(target=d.serve_forever, daemon=True)
    t.start()
    fc = None
    try:
        assert d.transport == "zmq" and d.port >= 40000 and d.server.endpoint == "tcp://127.0.0.1:%d" % d.port
        mgr = registry.redis_channelizer_manager(index=0, start_thread=False)
        t0 = time.time()
        while not mgr.channelizers:
            assert time.time() - t0 < 10, "the daemon never appeared in the registry"
            time.sleep(0.1)
            mgr.poll_once()
        rec = next(iter(mgr.channelizers.values()))
        assert rec["port"] == d.port and rec["index"] == 0 and rec["sources"] == [[FC0, int(FS)]]
        fc = FC.frontend_connector("test", mgr, heartbeat=False)              # transport_factory None: the zmq REQ link
        chan, port = fc.create_channel(CR, FC0 + 25000)
        assert chan and isinstance(port, str) and chan in d.tb.channels
        assert isinstance(fc._link.sock, zmq.Socket) and fc._link.sock.opts[zmq.RCVTIMEO] == 1000 and fc._link.sock.opts[zmq.LINGER] == 0
        # the data wire: a SUB socket on the channel's port, bare cf32 items (channel.py:36)
        sub = zmq.Context.instance().socket(zmq.SUB)
        sub.setsockopt(zmq.SUBSCRIBE, b"")
        sub.setsockopt(zmq.RCVHWM, 100000)
        sub.setsockopt(zmq.RCVTIMEO, 5000)
        sub.connect("tcp://127.0.0.1:%s" % port)
        buf = b""
        n = 4000
        while len(buf) < 8 * n:
            buf += sub.recv()
        assert len(buf) % 8 == 0
        got = np.frombuffer(buf[: 8 * n], dtype=np.complex64)
        sub.close()
        time.sleep(1.2)                                           # the record is refreshed once a second
        mgr.poll_once()
        start, decim = next(iter(mgr.channelizers.values()))["rcf_channel_starts"][chan]
        tile = sources.synthetic_tile(d.tb.realsources[0])
        need = start + (n + 60000) * decim
        x = np.tile(tile, need // len(tile) + 2)[start:need]
        want = G.xlating_fir_ccc(x, decim, G.channel_params(FS, CR)[1], 25000.0, FS)
        k0 = _align(got, want)
        ref = want[k0:k0 + n]
        assert np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2)) < 1e-6
        # heartbeat, drift report, then silence: the server forgets the client after 5 s (receiver.py:654-668) -- its clock
        # is moved instead of waiting
        assert fc.heartbeat_once() is True and fc.report_offset(0.25) is True
        cid = int(fc.my_client_id)
        d.server.client_hb[cid] -= 10
        t0 = time.time()
        while d.tb.channels[chan].in_use:
            assert time.time() - t0 < 5, "heartbeat expiry did not release the channel"
            time.sleep(0.02)
        assert fc.heartbeat_once() is False                        # 'fail,<cid>': the server forgot the client
        # a garbage request and an unknown verb: the REP socket answers every request (or it would wedge)
        raw = zmq.Context().socket(zmq.REQ)
        raw.setsockopt(zmq.RCVTIMEO, 2000)
        raw.connect("tcp://127.0.0.1:%d" % d.port)
        raw.send_string("create,not,a,number")
        assert raw.recv_string() in ("na", "na,not") or raw is None
        raw.send_string("quit,%d" % cid)
        assert raw.recv_string() == "quit,%d" % cid
        raw.close()
        verbs = [p.split(",")[0] for k, _, p in zmq.log if k == "req"]
        assert verbs[:2] == ["connect", "create"] and "hb" in verbs and "offset" in verbs and verbs[-1] == "quit"
    finally:
        if fc is not None:
            fc.exit()
        d.stop()
        t.join(timeout=10)
    assert not t.is_alive()
    assert d.port not in fake_zmq._bound                          # the REP socket is gone with the daemon ... not necessarily closed by zmq itself


def test_zmq_pub_factory_binds_one_pub_socket_per_port(fakes):
    from rcf import egress
    zmq, _ = fakes
    make = egress.zmq_pub_factory()
    a = make(51000)
    assert isinstance(a, zmq.Socket) and a.kind == zmq.PUB and a.port == 51000
    with pytest.raises(zmq.ZMQError):
        make(51000)
    a.close()
    make(51000).close()


def test_the_references_own_client_conversation_replays(fakes):
    """tests/golden/zmq_transcript.json: what the REFERENCE's frontend_connector.py (run in the build container against
    this channelizer over the same ZeroMQ stand-in: tests/golden/make_zmq_transcript.py) put on the wire and got back.
    A fresh daemon answers the same requests the same way (uuids / ports normalised) -- including the idle channel handed
    out again to the next create and the 'na' for a frequency outside every source."""
    import re
    zmq, _ = fakes
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "zmq_transcript.json")))
    d = frontend.Daemon(_config(), index=0, transport="zmq", registry="none", bind="127.0.0.1", frontend_factory=OracleFrontend)
    t = threading.Thread(target=d.serve_forever, daemon=True)
    t.start()
    uuid_re = re.compile(r"[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}")
    live = {}                                             # placeholder -> live value

    def norm(s):
        for ph, v in live.items():
            s = s.replace(v, ph)
        return s
    try:
        req = zmq.Context().socket(zmq.REQ)
        req.setsockopt(zmq.RCVTIMEO, 2000)
        req.connect("tcp://127.0.0.1:%d" % d.port)
        n_exchanges = 0
        for c in gold["calls"]:
            wire = c["wire"]
            for (k0, q), (k1, want) in zip(wire[0::2], wire[1::2]):
                assert (k0, k1) == ("req", "rep")
                for ph, v in live.items():
                    q = q.replace(ph, v)
                req.send_string(q)
                got = req.recv_string()
                m = re.match(r"^create,(%s),(\d+)$" % uuid_re.pattern, got)
                if m:
                    live.setdefault("<uuid%d>" % sum(1 for p in live if p.startswith("<uuid")), m.group(1)) if m.group(1) not in live.values() else None
                    live.setdefault("<port%d>" % sum(1 for p in live if p.startswith("<port")), m.group(2)) if m.group(2) not in live.values() else None
                assert norm(got) == want, (q, got, want)
                n_exchanges += 1
        assert n_exchanges >= 9
        for q in gold["heartbeat_exchange"]:              # its heartbeat thread's one request
            req.send_string(q)
            assert req.recv_string() in (q, q.replace("hb", "fail"))
        req.close()
    finally:
        d.stop()
        t.join(timeout=10)
