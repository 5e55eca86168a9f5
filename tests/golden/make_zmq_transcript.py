#!/usr/bin/env python3
"""Generate tests/golden/zmq_transcript.json: the REFERENCE's own frontend_connector.py driving THIS repository's
channelizer process over a ZeroMQ bus.

Runs only in the build container (needs /root/reference; nothing of it travels: the JSON holds the request / reply
strings that crossed the REQ / REP pair and what the reference's methods returned).  pyzmq is not installable here, so the
bus is tests/fake_zmq.py -- injected as `zmq` for BOTH sides: the reference's client code (frontend_connector.py:41-100
builds a zmq.Context, a REQ socket with its 1 s timeouts, connects, send_string / recv_string) and this repository's
rcf.protocol.FrontendServer.serve_zmq + rcf.egress.zmq_pub_factory.  The front-end behind the daemon is the oracle's
arithmetic (tests/test_daemon.py: OracleFrontend).  tests/test_zmq_redis_branches.py replays the requests against a
fresh daemon and compares the replies (uuids / ports / client ids normalised)."""
import json
import os
import re
import sys
import threading
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import fake_redis  # noqa: E402
import fake_zmq  # noqa: E402
sys.modules["zmq"] = fake_zmq
sys.modules["redis"] = fake_redis
config = types.ModuleType("config")
config.rc_config = type("rc_config", (), {"redis_servers": [("127.0.0.1", 6379)]})
sys.modules["config"] = config
sys.path.insert(0, "/root/reference")
import frontend_connector as REF_FC  # noqa: E402  (the reference's)
from rcf import frontend  # noqa: E402
from test_daemon import CR, FC0, OracleFrontend, _config  # noqa: E402

UUID = re.compile(r"[0-9a-f]{8}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{4}-[0-9a-f]{12}")


def main():
    d = frontend.Daemon(_config(), index=0, transport="zmq", registry="none", bind="127.0.0.1", frontend_factory=OracleFrontend)
    t = threading.Thread(target=d.serve_forever, daemon=True)
    t.start()

    class RCM:
        def get_channelizer_for_frequency(self, f):
            return ("127.0.0.1", d.port)
    calls = []
    fc = REF_FC.frontend_connector("parent-uuid", RCM())
    try:
        def call(name, *args):
            n0 = len(fake_zmq.log)
            ret = getattr(fc, name)(*args)
            time.sleep(0.05)
            calls.append({"call": name, "args": list(args), "returns": list(ret) if isinstance(ret, tuple) else ret,
                          "wire": [[k, p] for k, _, p in fake_zmq.log[n0:]]})
            return ret
        chan, port = call("create_channel", CR, FC0 + 25000)
        assert chan and chan in d.tb.channels, "the reference's client did not get a channel out of this channelizer"
        time.sleep(0.6)                                   # its heartbeat thread beats every 0.25 s meanwhile
        call("report_offset", 0.25)
        call("report_offset", 2.0)
        call("release_channel")
        call("create_channel", CR, FC0 - 12500)           # the idle channel is re-used (receiver.py:311-319)
        call("create_channel", CR, 100)                   # outside every source: 'na'
        call("release_channel")
        call("exit")
    finally:
        try:
            fc.continue_running = False
        except Exception:
            pass
        d.stop()
        t.join(timeout=10)
    ids, ports = {}, {}

    def norm(s):
        s = UUID.sub(lambda m: ids.setdefault(m.group(0), "<uuid%d>" % len(ids)), s)
        m = re.match(r"^(create,<uuid\d+>,)(\d+)$", s)
        if m:
            s = m.group(1) + ports.setdefault(m.group(2), "<port%d>" % len(ports))
        return s

    def norm_ret(r):
        if isinstance(r, list):
            return [norm_ret(x) for x in r]
        if isinstance(r, str):
            r = norm(r)
            return ports.get(r, r)
        return r
    out = {"what": "reference frontend_connector.py (imported from /root/reference with tests/fake_zmq.py as zmq) against "
                   "rcf.frontend.Daemon(transport='zmq'); uuids / ports normalised; heartbeats ('hb,<cid>' -> 'hb,<cid>') "
                   "of its 0.25 s thread are interleaved where they fell and are kept apart",
           "calls": [], "heartbeats": 0}
    for c in calls:
        wire = [[k, norm(p)] for k, p in c["wire"]]
        hb = [w for w in wire if w[1].startswith("hb,")]
        out["heartbeats"] += len(hb) // 2
        out["calls"].append({"call": c["call"], "args": c["args"], "returns": norm_ret(c["returns"]),
                             "wire": [w for w in wire if not w[1].startswith("hb,")]})
    hb_all = [p for k, _, p in fake_zmq.log if p.startswith("hb,")]
    out["heartbeats_total"] = len(hb_all) // 2
    out["heartbeat_exchange"] = sorted(set(hb_all))
    json.dump(out, open(os.path.join(HERE, "zmq_transcript.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
