#!/usr/bin/env python3
"""Generate tests/golden/protocol.json by importing the REFERENCE's pure-Python protocol code.

Runs only in the build container (needs /root/reference); the JSON it writes is data (request
strings, parsed replies, selection outcomes) -- no reference source travels.

  * /root/reference/frontend_connector.py is imported with a stub `zmq` module whose REQ socket
    records every request and plays back scripted replies.
  * /root/reference/redis_channelizer_manager.py is imported with stub `redis` + `config` modules;
    `get_channelizer_for_frequency` is exercised on hand-built channelizer tables.
"""
import json
import os
import random
import sys
import time
import types

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "protocol.json")

# ---------------------------------------------------------------- stub zmq
sent = []
replies = []


class _Sock:
    def setsockopt(self, *a): pass
    def getsockopt(self, *a): return 1000
    def connect(self, addr): sent.append(("CONNECT", addr))
    def close(self): pass
    def send_string(self, s): sent.append(("REQ", s))
    def recv_string(self):
        return replies.pop(0)


class _Ctx:
    def socket(self, kind): return _Sock()
    def term(self): pass
    def destroy(self): pass


zmq = types.ModuleType("zmq")
zmq.Context = _Ctx
zmq.REQ, zmq.RCVTIMEO, zmq.SNDTIMEO, zmq.LINGER = 3, 27, 28, 17
sys.modules["zmq"] = zmq

# ---------------------------------------------------------------- stub redis / config
redis = types.ModuleType("redis")
class _R:
    def __init__(self, *a, **k): pass
    def smembers(self, k): return set()
    def get(self, k): return None
redis.StrictRedis = _R
sys.modules["redis"] = redis
config = types.ModuleType("config")
class rc_config:
    redis_servers = [("127.0.0.1", 6379)]
config.rc_config = rc_config
sys.modules["config"] = config

sys.path.insert(0, REF)
import frontend_connector as FC          # noqa: E402
# the 0.25 s heartbeat thread (frontend_connector.py:197-229) would race the scripted replies; its
# one request ('hb,<cid>', :210) is captured explicitly through the same send() path instead.
_real_connection_handler = FC.frontend_connector.connection_handler
FC.frontend_connector.connection_handler = lambda self: None
import redis_channelizer_manager as RCM  # noqa: E402


class FakeRCM:
    def get_channelizer_for_frequency(self, f):
        return ("10.0.0.5", 4242)


golden = {"connector": [], "rcm": []}


def run_case(name, script):
    """script: list of (method, args, [replies...])"""
    global sent, replies
    fc = FC.frontend_connector("parent-uuid", FakeRCM())
    steps = []
    for method, args, reps in script:
        sent.clear()
        replies[:] = list(reps)
        ret = getattr(fc, method)(*args)
        steps.append({"call": method, "args": list(args), "replies": list(reps),
                      "requests": [s for k, s in sent if k == "REQ"],
                      "connects": [s for k, s in sent if k == "CONNECT"],
                      "returns": list(ret) if isinstance(ret, tuple) else ret,
                      "host": fc.host})
    golden["connector"].append({"name": name, "steps": steps})


run_case("create_release", [
    ("create_channel", (12500, 855000000), ["connect,7", "create,abc-uuid,12345"]),
    ("report_offset", (0.25,), ["offset,7"]),
    ("release_channel", (), ["release,abc-uuid"]),
])
run_case("heartbeat", [
    ("create_channel", (12500, 855000000), ["connect,7", "create,abc-uuid,12345"]),
    ("send", ("hb,7",), ["hb,7"]),
    ("send", ("hb,7",), ["fail,7"]),
    ("send", ("quit,7",), ["quit,7"]),
])
run_case("create_refused", [
    ("create_channel", (12500, 100), ["connect,3", "na,100"]),
    ("release_channel", (), []),
    ("report_offset", (1.5,), []),
])
run_case("release_refused", [
    ("create_channel", (25000, 851000000), ["connect,0", "create,u-1,10001"]),
    ("release_channel", (), ["na,u-1"]),
])
run_case("scan_mode", [
    ("create_channel", (12500, 5000), ["connect,1", "create,u-2,20002"]),
    ("scan_mode_set_freq", (770000000,), ["success"]),
])

# seeded random scripts: every call in every order, replies that succeed, refuse (na / fail), or are malformed, so that
# the mirror's state machine (connect on first use, host kept or dropped, what is returned on a refusal) is pinned on
# paths nobody wrote a case for
rs = random.Random(20260929)
for case in range(80):
    script = []
    for _ in range(rs.randint(1, 7)):
        u = rs.random()
        if u < 0.35:
            rate, freq = rs.choice([12500, 25000, 6250]), rs.choice([855000000, 851012500, 100, 5000, 770000000])
            kind = rs.random()
            if kind < 0.6:
                reps = ["connect,%d" % rs.randint(0, 99), "create,u-%d,%d" % (rs.randint(0, 999), rs.randint(10000, 60000))]
            elif kind < 0.8:
                reps = ["connect,%d" % rs.randint(0, 99), "na,%d" % freq]
            else:
                reps = ["connect,%d" % rs.randint(0, 99), rs.choice(["fail,1", "create,only-two", "garbage"])]
            script.append(("create_channel", (rate, freq), reps))
        elif u < 0.55:
            script.append(("release_channel", (), [rs.choice(["release,u-1", "na,u-1", "fail,0"])]))
        elif u < 0.75:
            # (any other reply leaves the reference's report_offset holding its thread_lock for good: frontend_connector.py
            #  falls off the end of the if / elif chain without releasing it -- the next call would never return)
            script.append(("report_offset", (rs.choice([0.1, 0.25, 0.75, 1.5, -2.0, 0.0]),), [rs.choice(["offset,7", "na,7"])]))
        elif u < 0.9:
            script.append(("send", (rs.choice(["hb,7", "quit,7"]),), [rs.choice(["hb,7", "fail,7", "quit,7"])]))
        else:
            script.append(("scan_mode_set_freq", (rs.choice([770000000, 855000000]),), [rs.choice(["success", "fail"])]))
    try:
        run_case("random_%02d" % case, script)
    except Exception as e:          # a script the reference itself cannot get through (e.g. a reply it cannot parse): not a golden
        print("skipped random_%02d: %s: %s" % (case, type(e).__name__, e))

# ---------------------------------------------------------------- the heartbeat loop (frontend_connector.py:197-229)
# run synchronously: its sleeps are counted instead of slept, the loop is stopped after a scripted number of beats.  Each
# beat is answered 'hb', 'fail' (the channelizer forgot the client: teardown, init, connect again) or not at all.
golden["heartbeat"] = []
for case in range(30):
    fc = FC.frontend_connector("parent-uuid", FakeRCM())
    sent.clear()
    replies[:] = ["connect,7", "create,u-1,12345"]
    fc.create_channel(12500, 855000000)
    beats = [rs.choice(["ok", "ok", "fail", "silent"]) for _ in range(rs.randint(1, 8))]
    script = []
    cid = 7
    for b in beats:
        if b == "ok":
            script.append("hb,%d" % cid)
        elif b == "fail":
            cid += 1
            script += ["fail,0", "connect,%d" % cid]
        else:
            cid += 1
            script.append(None)                                   # no reply: recv raises, five tries, then reconnect
            script.append("connect,%d" % cid)
    script.append("quit,%d" % cid)
    sent.clear()

    # a silent beat makes send() try to receive five times: give it five timeouts
    expanded = []
    for v in script:
        expanded += [None] * 5 if v is None else [v]
    queue = list(expanded)

    def recv2():
        v = queue.pop(0)
        if v is None:
            raise Exception("timeout")
        return v
    _Sock.recv_string = lambda self: recv2()
    n = {"beats": 0}
    real_sleep = FC.time.sleep

    def fake_sleep(t):
        if t == 0.25:
            n["beats"] += 1
            if n["beats"] >= len(beats):
                fc.continue_running = False
    FC.time.sleep = fake_sleep
    try:
        _real_connection_handler(fc)
    finally:
        FC.time.sleep = real_sleep
        _Sock.recv_string = lambda self: replies.pop(0)
    golden["heartbeat"].append({"beats": beats, "replies": script,
                                "requests": [s2 for k, s2 in sent if k == "REQ"],
                                "connects": [s2 for k, s2 in sent if k == "CONNECT"],
                                "client_id_after": fc.my_client_id})

# ---------------------------------------------------------------- rcm selection rule
mgr = RCM.redis_channelizer_manager.__new__(RCM.redis_channelizer_manager)
import logging
mgr.log = logging.getLogger("x")
mgr.clients = ["dummy"]
tables = {
    "two_sources_nearest_wins": {
        "A": {"address": "10.0.0.1", "port": 5001, "sources": [[855000000, 2400000]]},
        "B": {"address": "10.0.0.2", "port": 5002, "sources": [[855900000, 2400000]]},
    },
    "edge_exclusive": {
        "A": {"address": "10.0.0.1", "port": 5001, "sources": [[855000000, 2400000]]},
    },
    "multi_source_channelizer": {
        "A": {"address": "10.0.0.1", "port": 5001,
              "sources": [[851000000, 2400000], [853000000, 2400000]]},
        "B": {"address": "10.0.0.2", "port": 5002, "sources": [[852900000, 10000000]]},
    },
}
queries = {
    "two_sources_nearest_wins": [855100000, 855500000, 856900000, 853700000, 860000000],
    "edge_exclusive": [856200000, 856199999, 853800000, 853800001],
    "multi_source_channelizer": [852000000, 853000001, 850000000, 857899999, 857900000],
}
for name, table in tables.items():
    mgr.channelizers = table
    for q in queries[name]:
        random.seed(0)
        got = mgr.get_channelizer_for_frequency(q)
        golden["rcm"].append({"table": name, "channelizers": table, "frequency": q,
                              "result": list(got)})

# seeded random tables and queries (overlapping sources, channelizers with several sources, queries on the band edges)
for t in range(40):
    table = {}
    for c in range(rs.randint(1, 5)):
        table["C%d" % c] = {"address": "10.0.%d.%d" % (t, c), "port": 5000 + c,
                            "sources": [[rs.choice([851000000, 853000000, 855000000, 855900000, 860000000]) + rs.randint(-3, 3) * 100000,
                                         rs.choice([2400000, 8000000, 10000000, 20000000])] for _ in range(rs.randint(1, 3))]}
    mgr.channelizers = table
    for _ in range(6):
        cf, bw = rs.choice([s2 for c in table.values() for s2 in c["sources"]])
        q = rs.choice([cf, cf + bw // 2, cf - bw // 2, cf + bw // 2 - 1, cf - bw // 2 + 1, cf + rs.randint(-bw, bw)])
        random.seed(0)
        try:
            got = mgr.get_channelizer_for_frequency(q)
        except Exception as e:
            got = ("EXC", type(e).__name__)
        golden["rcm"].append({"table": "random_%02d" % t, "channelizers": table, "frequency": q, "result": list(got)})

# ---------------------------------------------------------------- one pass of the manager's poll loop
# redis_channelizer_manager.manager_loop (:79-124) run ONCE (its sleep ends the loop) over seeded random registry
# contents: records of different ages around the 5 s expiry, with and without a device index, one key without a record.
# Written: the registry before, the index the manager filters on, the channelizers it ends up with and what it removed.
golden["rcm_poll"] = []
for case in range(40):
    now = 1000000.0
    table, kv = set(), {}
    for c in range(rs.randint(0, 6)):
        uid = "uuid-%d-%d" % (case, c)
        table.add(uid.encode())
        if rs.random() < 0.1:
            continue                                              # announced, record missing
        rec = {"address": "10.0.0.%d" % c, "port": 5000 + c, "sources": [[855000000 + c * 1000000, 2400000]],
               "current_time": now - rs.choice([0.0, 1.0, 4.9, 4.999, 5.0, 5.001, 6.0, 60.0])}
        if rs.random() < 0.7:
            rec["index"] = rs.choice([0, 1, 3])
        kv[uid] = json.dumps(rec)
    removed = []

    class _Poll:
        def smembers(self, k): return set(table)
        def get(self, k): return kv.get(k.decode() if isinstance(k, bytes) else k)
        def srem(self, k, v): removed.append(["srem", k, v])
        def delete(self, k): removed.append(["delete", k])

    m = RCM.redis_channelizer_manager.__new__(RCM.redis_channelizer_manager)
    m.log = logging.getLogger("x")
    m.clients = [_Poll()]
    m.index = rs.choice([None, None, 0, 1, 3])
    m.channelizers = {}
    m.continue_running = True
    real_time, real_sleep = RCM.time.time, RCM.time.sleep
    RCM.time.time = lambda: now
    RCM.time.sleep = lambda s: setattr(m, "continue_running", False)
    try:
        m.manager_loop()
    finally:
        RCM.time.time, RCM.time.sleep = real_time, real_sleep
    golden["rcm_poll"].append({"now": now, "index": m.index, "members": sorted(t.decode() for t in table), "records": kv,
                               "channelizers": m.channelizers, "removed": removed})

# ---------------------------------------------------------------- the registry record, as the reference publishes it
# rc_frontend/redis_channel_publisher.py run with a recording redis pipeline and a ZMQ socket stub: one pass of its
# publish loop, with and without a device index.  Volatile values (uuid, times, host, pid, address) are replaced by their
# type names; everything a reader keys on is kept.
sys.path.insert(0, os.path.join(REF, "rc_frontend"))
zmq.LAST_ENDPOINT = 32
ops = []


class _Pipe:
    def sadd(self, k, v): ops.append(["sadd", k, v])
    def set(self, k, v): ops.append(["set", k, v])
    def execute(self): ops.append(["execute"])


_R.pipeline = lambda self: _Pipe()


class _ZSock:
    def getsockopt(self, opt): return b"tcp://0.0.0.0:47123"


import redis_channel_publisher as RCP    # noqa: E402
golden["publisher"] = []
for index in (None, 3):
    del ops[:]
    srcs = {0: {"center_freq": 855050000, "samp_rate": 2400000}, 1: {"center_freq": 857000000.0, "samp_rate": 8000000}}
    pub = RCP.redis_channel_publisher(sources=srcs, channels={"a": 1, "b": 2}, zmq_socket=_ZSock(), index=index)
    t0 = time.time()
    while not any(o[0] == "execute" for o in ops) and time.time() - t0 < 5:
        time.sleep(0.05)
    pub.continue_running = False
    first = ops[:[o[0] for o in ops].index("execute") + 1]
    rec = json.loads([o for o in first if o[0] == "set"][0][2])
    volatile = ("instance_uuid", "start_time", "current_time", "hostname", "pid", "address")
    golden["publisher"].append({
        "index": index,
        "sources": {str(k): v for k, v in srcs.items()}, "n_channels": 2, "port": 47123,
        "ops": [[o[0]] + ([o[1]] if o[0] == "sadd" else []) for o in first],
        "key_is_uuid_in_both_ops": first[0][2] == first[1][1] == rec["instance_uuid"],
        "record": {k: (type(v).__name__ if k in volatile else v) for k, v in rec.items()},
    })

with open(OUT, "w") as f:
    json.dump(golden, f, indent=1, sort_keys=True)
print("wrote", OUT, len(golden["connector"]), "connector cases,", len(golden["rcm"]), "rcm queries")
os._exit(0)
