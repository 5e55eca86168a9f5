"""Drop-in for /root/reference/frontend_connector.py: the channel client backends use.

Same constructor, methods and return types (`create_channel` -> (channel_id, port) with `port` a
STRING as the reference returns it, `(False, False)` on failure; `release_channel`, `report_offset`,
`scan_mode_set_freq`, `exit`, attribute `.host`), same wire strings (tests/golden/protocol.json was
captured from the reference).  The socket is pluggable: pyzmq REQ when available, else any object
with send_string/recv_string (rcf.protocol.LoopbackTransport).
"""
from __future__ import annotations

import logging
import threading
import time
import uuid


class frontend_connector():
    def __init__(self, parent_instance_uuid, redis_channelizer_manager, transport_factory=None,
                 heartbeat=True):
        self.log = logging.getLogger('%s.frontend_connector' % (str(uuid.uuid4())))
        self.thread_lock = threading.Lock()
        self.send_lock = threading.Lock()
        self.continue_running = True
        self.redis_channelizer_manager = redis_channelizer_manager
        self.transport_factory = transport_factory
        self.channel_port = 0
        self.host = None
        self.context = None
        self.socket = None
        self.my_client_id = None
        self.channel_id = None
        self.frequency = None
        self.last_create_channel = None
        if heartbeat:
            t = threading.Thread(target=self.connection_handler, name='connection_handler')
            t.daemon = True
            t.start()

    # -- connection (frontend_connector.py:41-73)
    def connection_init(self, frequency):
        host, port = self.redis_channelizer_manager.get_channelizer_for_frequency(frequency)
        self.host = host
        if self.transport_factory is not None:
            self.context = object()
            self.socket = self.transport_factory(host, port)
        else:
            import zmq
            self.context = zmq.Context()
            self.socket = self.context.socket(zmq.REQ)
            self.socket.setsockopt(zmq.RCVTIMEO, 1000)
            self.socket.setsockopt(zmq.SNDTIMEO, 1000)
            self.socket.setsockopt(zmq.LINGER, 0)
            self.socket.connect("tcp://%s:%s" % (host, port))
        self.my_client_id = None
        self.channel_id = None
        self.channel_port = 0

    def connection_teardown(self):
        for fn in (lambda: self.socket.close(), lambda: self.context.term(), lambda: self.context.destroy()):
            try:
                fn()
            except Exception:
                pass

    # -- request/reply with 5 tries each way (frontend_connector.py:75-96)
    def send(self, data):
        tries = 0
        sent = False
        while tries < 5 and not sent:
            try:
                with self.send_lock:
                    self.socket.send_string(data)
                    sent = True
            except Exception as e:
                self.log.error('Exception in frontend_connector.send(): %s %s' % (type(e), e))
                tries += 1
        while tries < 5:
            try:
                with self.send_lock:
                    response = self.socket.recv_string()
                    return response.split(',')
            except Exception as e:
                self.log.error('Exception in frontend_connector.recv(): %s %s' % (type(e), e))
                tries += 1
        return None

    def connect(self):
        data = self.send('connect')
        if data is None:
            return None
        self.my_client_id = int(data[1])

    def scan_mode_set_freq(self, freq):
        with self.thread_lock:
            data = self.send('scan_mode_set_freq,%s' % (freq))
        # the reference compares the split LIST with the string 'success' (frontend_connector.py:122):
        # that is never equal, so it always returns False; kept bug-compatible
        if data == 'success':
            return True
        return False

    def create_channel(self, channel_rate, freq):
        self.last_create_channel = time.time()
        self.frequency = freq
        self.connection_init(freq)
        self.connect()
        with self.thread_lock:
            data = self.send('create,%s,%s,%s' % (self.my_client_id, channel_rate, freq))
            if data is None or data[0] == 'na':
                self.log.error('Failed to create channel')
                return False, False
            elif data[0] == 'create':
                self.channel_id = data[1]
                self.channel_port = data[2]
                return self.channel_id, self.channel_port
            return False, False

    def release_channel(self):
        with self.thread_lock:
            if self.channel_id is None:
                return False
            data = self.send('release,%s,%s' % (self.my_client_id, self.channel_id))
            self.frequency = None
            if data is None or data[0] == 'na':
                self.log.error('Failed to release channel, probably leaking channels')
                return False
            elif data[0] == 'release':
                channel_id = data[1]
                self.channel_id = None
                return channel_id
            return False

    def report_offset(self, offset):
        with self.thread_lock:
            if self.channel_id is None:
                return False
            data = self.send('offset,%s,%s,%s' % (self.my_client_id, self.channel_id, offset))
            if data is None or data[0] == 'na':
                self.log.error('Failed to set offset')
                return False
            elif data[0] == 'offset':
                return True

    def exit(self):
        self.continue_running = False

    # -- 0.25 s heartbeat, reconnect on failure (frontend_connector.py:197-229)
    def heartbeat_once(self):
        data = self.send('hb,%s' % self.my_client_id)
        if data is None or data[0] == 'fail':
            self.log.error('Failed to heartbeat')
            self.connection_teardown()
            self.connection_init(self.frequency)
            self.connect()
            return False
        return True

    def connection_handler(self):
        time.sleep(0.1)
        while self.continue_running:
            if self.context is None or self.host is None:
                time.sleep(0.01)
                continue
            with self.thread_lock:
                try:
                    self.heartbeat_once()
                except Exception as e:
                    self.log.error('Failed to heartbeat: %s' % e)
            time.sleep(0.25)
        with self.thread_lock:
            try:
                self.send('quit,%s' % self.my_client_id)
                self.socket.close()
            except Exception:
                pass
