"""An in-process stand-in for redis-py's StrictRedis with the six commands the channelizer registry uses
(rc_frontend/redis_channel_publisher.py:63-90: SADD, SET; redis_channelizer_manager.py:78-124: SMEMBERS, GET, SREM, DELETE)
and redis-py's return TYPES -- members and values come back as bytes, SADD / SREM / DELETE return counts, SET returns True,
GET of a missing key None -- so that rcf.registry's DEFAULT clients (redis.StrictRedis(host='127.0.0.1', port=6379, db=0))
run in the CPU suite (redis-py and a server are not installable here).  One store per (host, port, db),
shared by every client object of the process, as a server would; `reset()` clears them."""
import threading

_lock = threading.Lock()
_stores = {}
calls = []                    # (command, args...) in order, for transcripts


def reset():
    with _lock:
        _stores.clear()
        del calls[:]


def _b(v):
    if isinstance(v, bytes):
        return v
    if isinstance(v, (int, float)):
        v = repr(v) if isinstance(v, float) else str(v)
    return str(v).encode("utf-8")


class ConnectionError(Exception):
    pass


class StrictRedis:
    def __init__(self, host="localhost", port=6379, db=0, **kw):
        key = (str(host), int(port), int(db))
        with _lock:
            self._s = _stores.setdefault(key, {"kv": {}, "sets": {}})
        self.connection_args = key

    def ping(self):
        return True

    def sadd(self, name, *values):
        with _lock:
            calls.append(("SADD", name) + tuple(values))
            st = self._s["sets"].setdefault(_b(name), set())
            n0 = len(st)
            st.update(_b(v) for v in values)
            return len(st) - n0

    def srem(self, name, *values):
        with _lock:
            calls.append(("SREM", name) + tuple(values))
            st = self._s["sets"].get(_b(name), set())
            n0 = len(st)
            for v in values:
                st.discard(_b(v))
            return n0 - len(st)

    def smembers(self, name):
        with _lock:
            return set(self._s["sets"].get(_b(name), set()))

    def set(self, name, value, **kw):
        with _lock:
            calls.append(("SET", name, value))
            self._s["kv"][_b(name)] = _b(value)
            return True

    def get(self, name):
        with _lock:
            return self._s["kv"].get(_b(name))

    def delete(self, *names):
        with _lock:
            calls.append(("DELETE",) + tuple(names))
            n = 0
            for k in names:
                n += 1 if self._s["kv"].pop(_b(k), None) is not None else 0
            return n


Redis = StrictRedis
