"""An in-process stand-in for pyzmq with the semantics the channelizer's ZeroMQ branches rely on -- so that every line of
rcf.protocol.FrontendServer.serve_zmq, rcf.egress.zmq_pub_factory, rcf.frontend_connector's default REQ link and
rcf.frontend.Daemon(transport='zmq') runs in the CPU suite (pyzmq is not installable here).

What is modelled, because the code under test depends on it:
  * REQ / REP lock step: a REQ socket must recv() the reply before it may send() again (ZMQError EFSM), a REP socket must
    send() a reply before the next recv(); a request to an endpoint nobody is bound to waits (zmq queues it) and the recv()
    times out with zmq.Again after RCVTIMEO -- what frontend_connector's 1 s timeouts and 5 retries are written against
  * recv_string(flags=NOBLOCK) raises zmq.Again when nothing is queued (the reference's main loop, receiver.py:686-690)
  * bind('tcp://0.0.0.0:0') picks an ephemeral port, getsockopt(LAST_ENDPOINT) names it (receiver.py:46)
  * bind() on a port that is already bound raises ZMQError (EADDRINUSE): the receiver's port-retry loop
  * PUB / SUB: no back-pressure -- a message is DROPPED for a subscriber whose queue is at its high-water mark, subscribers
    that connect later miss what was sent before; prefix subscription (b'' = everything)
  * Context() / Context.instance(), socket.close(), context.term() / destroy(), LINGER / RCVTIMEO / SNDTIMEO options
Everything lives in one process-wide bus keyed by TCP port; `reset()` clears it between tests.
"""
import collections
import threading
import time

REQ, REP, PUB, SUB = 3, 4, 1, 2
RCVTIMEO, SNDTIMEO, LINGER, SUBSCRIBE, UNSUBSCRIBE, LAST_ENDPOINT, SNDHWM, RCVHWM = 27, 28, 17, 6, 7, 32, 23, 24
NOBLOCK = DONTWAIT = 1
EAGAIN, EFSM, EADDRINUSE = 11, 156384763, 98


class ZMQError(Exception):
    def __init__(self, errno=None, msg=None):
        super().__init__(msg or "zmq error %s" % errno)
        self.errno = errno


class Again(ZMQError):
    def __init__(self, msg="Resource temporarily unavailable"):
        super().__init__(EAGAIN, msg)


_lock = threading.RLock()
_cond = threading.Condition(_lock)
_bound = {}                     # port -> socket (REP or PUB)
_next_port = [40000]
log = []                        # (kind, port, payload): every request / reply that crossed the bus, for transcripts


def reset():
    with _lock:
        _bound.clear()
        _next_port[0] = 40000
        del log[:]


def _port_of(addr):
    if not addr.startswith("tcp://"):
        raise ZMQError(msg="fake zmq speaks tcp:// only, not %r" % addr)
    return int(addr.rsplit(":", 1)[1])


class Socket:
    def __init__(self, ctx, kind):
        self.ctx, self.kind, self.closed = ctx, kind, False
        self.opts = {RCVTIMEO: -1, SNDTIMEO: -1, LINGER: -1, SNDHWM: 1000, RCVHWM: 1000}
        self.inbox = collections.deque()        # REP: (req_socket, text); REQ: reply text; SUB: payload bytes
        self.port = None                        # bound port (REP / PUB)
        self.peer_port = None                   # connected port (REQ / SUB)
        self.endpoint = b""
        self.subs = []                          # PUB: connected SUB sockets; SUB: subscription prefixes
        self.awaiting_reply = False             # REQ: sent, not yet received
        self.reply_to = None                    # REP: the REQ socket the next send answers

    # ---- options
    def setsockopt(self, opt, val):
        if opt == SUBSCRIBE:
            self.subs.append(bytes(val))
        elif opt == UNSUBSCRIBE:
            if bytes(val) in self.subs:
                self.subs.remove(bytes(val))
        else:
            self.opts[opt] = val

    def getsockopt(self, opt):
        if opt == LAST_ENDPOINT:
            return self.endpoint
        return self.opts.get(opt, 0)

    # ---- wiring
    def bind(self, addr):
        host = addr[len("tcp://"):].rsplit(":", 1)[0]
        port = _port_of(addr)
        with _lock:
            if port == 0:
                while _next_port[0] in _bound:
                    _next_port[0] += 1
                port = _next_port[0]
                _next_port[0] += 1
            if port in _bound:
                raise ZMQError(EADDRINUSE, "Address already in use")
            _bound[port] = self
            self.port = port
            self.endpoint = ("tcp://%s:%d" % (host, port)).encode("utf-8")
            _cond.notify_all()

    def connect(self, addr):
        self.peer_port = _port_of(addr)
        if self.kind == SUB:
            with _lock:
                pub = _bound.get(self.peer_port)
                if pub is not None and pub.kind == PUB and self not in pub.subs:
                    pub.subs.append(self)

    def close(self, linger=None):
        with _lock:
            self.closed = True
            if self.port is not None and _bound.get(self.port) is self:
                del _bound[self.port]
            if self.kind == SUB and self.peer_port in _bound and self in _bound[self.peer_port].subs:
                _bound[self.peer_port].subs.remove(self)
            _cond.notify_all()

    # ---- helpers
    def _wait(self, ready, timeout_ms):
        """under _lock: wait until ready() or the timeout; True when ready"""
        deadline = None if timeout_ms is None or timeout_ms < 0 else time.monotonic() + timeout_ms / 1000.0
        while not ready():
            if self.closed:
                raise ZMQError(msg="socket closed")
            left = None if deadline is None else deadline - time.monotonic()
            if left is not None and left <= 0:
                return False
            _cond.wait(0.05 if left is None else min(left, 0.05))
        return True

    # ---- REQ / REP
    def send_string(self, s, flags=0):
        self.send(s.encode("utf-8"), flags)

    def recv_string(self, flags=0):
        return self.recv(flags).decode("utf-8")

    def send(self, data, flags=0):
        data = bytes(data)
        with _lock:
            if self.closed:
                raise ZMQError(msg="socket closed")
            if self.kind == REQ:
                if self.awaiting_reply:
                    raise ZMQError(EFSM, "Operation cannot be accomplished in current state")
                # zmq queues a request until a peer is there; the REQ end only notices at recv() (RCVTIMEO)
                self.awaiting_reply = True
                self._pending = data

                def deliver():
                    rep = _bound.get(self.peer_port)
                    if rep is None or rep.kind != REP or rep.closed:
                        return False
                    rep.inbox.append((self, data))
                    log.append(("req", self.peer_port, data.decode("utf-8", "replace")))
                    self._pending = None
                    _cond.notify_all()
                    return True
                self._deliver = deliver
                deliver()
            elif self.kind == REP:
                if self.reply_to is None:
                    raise ZMQError(EFSM, "Operation cannot be accomplished in current state")
                peer, self.reply_to = self.reply_to, None
                log.append(("rep", self.port, data.decode("utf-8", "replace")))
                if not peer.closed:
                    peer.inbox.append(data)
                _cond.notify_all()
            elif self.kind == PUB:
                for sub in list(self.subs):
                    if sub.closed:
                        continue
                    if not any(data.startswith(p) for p in sub.subs):
                        continue
                    if len(sub.inbox) >= max(1, sub.opts.get(RCVHWM, 1000)):
                        continue                              # at the high-water mark: dropped, no back-pressure
                    sub.inbox.append(data)
                _cond.notify_all()
            else:
                raise ZMQError(msg="a SUB socket cannot send")

    def recv(self, flags=0):
        with _lock:
            if self.closed:
                raise ZMQError(msg="socket closed")
            timeout = 0 if flags & NOBLOCK else self.opts.get(RCVTIMEO, -1)
            if self.kind == REQ:
                if not self.awaiting_reply:
                    raise ZMQError(EFSM, "Operation cannot be accomplished in current state")

                def ready():
                    if getattr(self, "_pending", None) is not None:
                        self._deliver()                       # the peer may have bound meanwhile
                    return bool(self.inbox)
                if not self._wait(ready, timeout):
                    raise Again()
                self.awaiting_reply = False
                return self.inbox.popleft()
            if self.kind == REP:
                if self.reply_to is not None:
                    raise ZMQError(EFSM, "Operation cannot be accomplished in current state")
                if not self._wait(lambda: bool(self.inbox), timeout):
                    raise Again()
                self.reply_to, data = self.inbox.popleft()
                return data
            if self.kind == SUB:
                if not self._wait(lambda: bool(self.inbox), timeout):
                    raise Again()
                return self.inbox.popleft()
            raise ZMQError(msg="a PUB socket cannot receive")


class Context:
    _instance = None

    def __init__(self, io_threads=1):
        self.sockets = []

    @classmethod
    def instance(cls):
        if cls._instance is None:
            cls._instance = cls()
        return cls._instance

    def socket(self, kind):
        s = Socket(self, kind)
        self.sockets.append(s)
        return s

    def term(self):
        for s in self.sockets:
            if not s.closed:
                s.close()

    def destroy(self, linger=None):
        self.term()
