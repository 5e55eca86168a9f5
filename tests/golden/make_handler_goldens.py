#!/usr/bin/env python3
"""Generate tests/golden/handler.json by RUNNING the reference's own REP handler.

rc_frontend/receiver.py defines `handler(msg, tb)` inside its `if __name__ == '__main__':` block (:503-614), behind
GNU Radio imports -- it cannot be imported.  This script (build container only: it needs /root/reference) parses the
file with `ast`, compiles that one function definition as it stands, gives it the globals it expects (`clients`,
`client_hb`, `client_num`, `log`, `time`) and a recording stand-in for the top block, and drives it with seeded random
sessions.  What is written is DATA: the messages, the replies, the calls the handler made on `tb` and the client
tables after every message -- no reference source travels.
"""
import ast
import json
import logging
import os
import random
import types

REF = "/root/reference/rc_frontend/receiver.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "handler.json")

tree = ast.parse(open(REF).read())
fn = None
for node in ast.walk(tree):
    if isinstance(node, ast.FunctionDef) and node.name == "handler":
        fn = node
assert fn is not None, "no handler() in the reference"
mod = ast.Module(body=[fn], type_ignores=[])
code = compile(mod, REF, "exec")


class Clock:
    """time.time() that advances by a scripted step per call: heartbeat stamps become reproducible"""
    def __init__(self):
        self.t = 1000.0

    def time(self):
        self.t += 0.25
        return self.t


class FakeTB:
    """the surface of receiver the handler touches; every call recorded, results scripted"""
    def __init__(self, rs):
        self.rs = rs
        self.calls = []
        self.outcomes = []
        self.n = 0
        self.channels = {}                       # the handler logs len(tb.channels)
        calls = self.calls

        class _Source:                           # scan_mode_set_freq goes to the SDR block itself (receiver.py:556-566)
            def set_center_freq(self, freq, chan):
                calls.append(["scan_mode_set_freq", freq])

        self.realsources = {0: {"block": _Source()}}

    def connect_channel(self, channel_rate, freq):
        self.calls.append(["connect_channel", channel_rate, freq])
        ok = self.rs.random() >= 0.12
        self.outcomes.append(ok)                 # written into the golden: the replay scripts its top block with it
        if not ok:
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


def session(seed):
    rs = random.Random(seed)
    clock = Clock()
    g = {"clients": {}, "client_hb": {}, "client_num": 0, "log": logging.getLogger("golden"),
         "time": types.SimpleNamespace(time=clock.time), "__builtins__": __builtins__}
    exec(code, g)
    handler = g["handler"]
    tb = FakeTB(rs)
    known_clients, known_blocks = [], []
    steps = []
    for _ in range(rs.randint(3, 25)):
        u = rs.random()
        if u < 0.2 or not known_clients:
            msg = "connect"
        elif u < 0.5:
            c = rs.choice(known_clients + [97])                      # 97: a client that never connected
            msg = "create,%s,%s,%s" % (c, rs.choice([12500, 25000]), rs.choice([855000000, 851012500, 5000, 100]))
        elif u < 0.65:
            c = rs.choice(known_clients + [97])
            b = rs.choice(known_blocks + ["blk-none"]) if known_blocks else "blk-none"
            msg = "release,%s,%s" % (c, b)
        elif u < 0.75:
            msg = "hb,%s" % rs.choice(known_clients + [97])
        elif u < 0.85:
            c = rs.choice(known_clients)
            b = rs.choice(known_blocks + ["blk-none"]) if known_blocks else "blk-none"
            msg = "offset,%s,%s,%s" % (c, b, rs.choice([0.25, -1.5, 0.0, 2.0]))
        elif u < 0.92:
            msg = "scan_mode_set_freq,%s" % rs.choice([770000000, 855000000])
        elif u < 0.97:
            msg = "quit,%s" % rs.choice(known_clients + [97])
        else:
            msg = rs.choice(["release,1", "hb,x", "bogus,1,2", "create,1,12500", " connect \n"])
        del tb.calls[:]
        del tb.outcomes[:]
        try:
            reply = handler(msg, tb)
            exc = None
        except Exception as e:                                       # the reference lets some malformed messages raise
            reply, exc = None, type(e).__name__
        if isinstance(reply, str) and reply.startswith("connect,"):
            known_clients.append(int(reply.split(",")[1]))
        if isinstance(reply, str) and reply.startswith("create,"):
            known_blocks.append(reply.split(",")[1])
        steps.append({"msg": msg, "reply": reply, "raises": exc, "calls": [list(c) for c in tb.calls],
                      "connect_ok": list(tb.outcomes),
                      "clients": {str(k): list(v) for k, v in g["clients"].items()},
                      "client_hb": sorted(str(k) for k in g["client_hb"]),
                      "client_num": g["client_num"]})
    return {"seed": seed, "steps": steps}


# ---------------------------------------------------------------- the main loop's housekeeping (receiver.py:620-680)
# The `while 1:` statement under __main__ -- 10 s status, idle-channel sweep, 5 s client-heartbeat expiry, then a
# non-blocking receive -- lifted out the same way and run for ONE turn at a scripted time (the receive raises zmq.Again,
# the sleep that follows ends the loop) over seeded random client tables and channel tables.
loop = None
for node in ast.walk(tree):
    if isinstance(node, ast.While) and isinstance(node.test, ast.Constant) and node.test.value == 1:
        loop = node
assert loop is not None, "no main loop in the reference"
loop_code = compile(ast.Module(body=[loop], type_ignores=[]), REF, "exec")


class _Stop(Exception):
    pass


def tick_case(seed):
    rs = random.Random(10000 + seed)
    now = 5000.0 + rs.random() * 100

    class Again(Exception):
        pass

    released, destroyed = [], []

    class Chan:
        def __init__(self, bid, close_time):
            self.block_id, self.channel_close_time = bid, close_time

        def destroy(self):
            destroyed.append(self.block_id)

    class TB:
        pass
    tb = TB()
    tb.channel_idle_timeout = 10
    tb.last_channel_cleanup = now - rs.choice([0.0, 5.0, 19.9, 20.1, 45.0])
    tb.channels = {}
    for i in range(rs.randint(0, 6)):
        bid = "blk-%d" % i
        tb.channels[bid] = Chan(bid, rs.choice([0, 0, now - 1.0, now - 9.9, now - 10.1, now - 60.0]))
    tb.release_channel = lambda bid: released.append(bid) or True
    clients, client_hb = {}, {}
    for c in range(rs.randint(0, 5)):
        if rs.random() < 0.9:
            clients[c] = ["blk-%d" % rs.randint(0, 6) for _ in range(rs.randint(0, 3))]
        if rs.random() < 0.9:
            client_hb[c] = now - rs.choice([0.0, 1.0, 4.9, 5.1, 30.0])
    before = {"clients": {str(k): list(v) for k, v in clients.items()}, "client_hb": {str(k): v for k, v in client_hb.items()},
              "channels": {k: v.channel_close_time for k, v in tb.channels.items()},
              "last_channel_cleanup": tb.last_channel_cleanup, "last_status": None}
    last_status = now - rs.choice([0.0, 9.9, 10.1, 100.0])
    before["last_status"] = last_status

    def sleep(t):
        raise _Stop()

    class Sock:
        def recv_string(self, flags=0):
            raise Again()

    g = {"clients": clients, "client_hb": client_hb, "tb": tb, "log": logging.getLogger("golden"),
         "time": types.SimpleNamespace(time=lambda: now, sleep=sleep), "zmq": types.SimpleNamespace(Again=Again, NOBLOCK=1),
         "socket": Sock(), "last_status": last_status, "start_time": now - 1000.0, "handler": None,
         "__builtins__": __builtins__}
    raised = None
    try:
        exec(loop_code, g)
    except _Stop:
        pass
    except Exception as e:                               # the reference's loop lets a KeyError out (client_hb without clients)
        raised = type(e).__name__
    return {"seed": seed, "now": now, "before": before, "raises": raised, "released": released, "destroyed": destroyed,
            "after": {"clients": {str(k): list(v) for k, v in clients.items()}, "client_hb": sorted(str(k) for k in client_hb),
                      "channels": sorted(tb.channels), "last_channel_cleanup": tb.last_channel_cleanup,
                      "last_status": g["last_status"]}}


golden = {"sessions": [session(s) for s in range(120)], "ticks": [tick_case(s) for s in range(150)]}
with open(OUT, "w") as f:
    json.dump(golden, f, indent=0, sort_keys=True)
print("wrote", OUT, len(golden["sessions"]), "sessions,", sum(len(s["steps"]) for s in golden["sessions"]), "messages")
