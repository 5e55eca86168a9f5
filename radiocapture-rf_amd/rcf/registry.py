"""Channelizer registry: /root/reference/rc_frontend/redis_channel_publisher.py (publisher) and
/root/reference/redis_channelizer_manager.py (reader), with the Redis client injected (redis-py when
installed; any object with sadd/set/smembers/get/srem/delete otherwise -- tests use a dict-backed fake).
"""
from __future__ import annotations

import json
import os
import random
import socket
import threading
import time
import uuid


class redis_channel_publisher():
    """Every 1 s: SADD channelizers <uuid>; SET <uuid> json{...} (redis_channel_publisher.py:59-93)."""

    def __init__(self, sources=None, channels=None, port=None, index=None, client=None, address=None,
                 extra=None, start_thread=True, health=None):
        if sources is None:
            raise Exception('Sources must be provided at initialization')
        if channels is None:
            raise Exception('Channels must be provided at initialization')
        if port is None:
            raise Exception('The control port (the reference reads it from the ZMQ socket) must be provided')
        self.start_time = time.time()
        self.sources = sources
        self.channels = channels
        self.port = port
        self.index = index
        self.address = address
        self.extra = extra or (lambda: {})       # additive keys only: receiver.metrics (Msps in, channels, fault)
        # health() -> False (or raising) stops the heartbeat: the manager side expires the record after 5 s and
        # clients reconnect elsewhere ("GPU error => stop heartbeating"); receiver.healthy is the feed
        self.health = health
        self.instance_uuid = str(uuid.uuid4())
        if client is None:
            import redis
            client = redis.StrictRedis(host='127.0.0.1', port=6379, db=0)   # publisher.py:27,31
        self.client = client
        self.continue_running = True
        if start_thread:
            t = threading.Thread(target=self.publish_loop)
            t.daemon = True
            t.start()

    def build(self, now=None):
        hostname = socket.gethostname()
        try:
            address = self.address or socket.gethostbyname(hostname)
        except Exception:
            address = '127.0.0.1'
        publish_data = {
            'instance_uuid': self.instance_uuid,
            'start_time': self.start_time,
            'current_time': time.time() if now is None else now,
            'hostname': hostname,
            'pid': os.getpid(),
            'address': address,
            'port': self.port,
            'channel_count': len(self.channels),
            'source_count': len(self.sources),
            'sources': [],
        }
        if self.index is not None:
            publish_data['index'] = self.index
        for source_id in self.sources:
            source = self.sources[source_id]
            publish_data['sources'].append((source['center_freq'], source['samp_rate']))
        publish_data.update(self.extra())
        return publish_data

    def is_healthy(self):
        if self.health is None:
            return True
        try:
            return bool(self.health())
        except Exception:
            return False

    def publish_once(self, now=None):
        if not self.is_healthy():
            return None
        data = self.build(now)
        self.client.sadd('channelizers', self.instance_uuid)
        self.client.set(self.instance_uuid, json.dumps(data))
        return data

    def publish_loop(self):
        time.sleep(0.5)
        while self.continue_running:
            try:
                self.publish_once()
            except Exception:
                pass
            time.sleep(1)


class redis_channelizer_manager():
    """Reader side: 0.5 s poll, 5 s expiry, nearest-centre selection (redis_channelizer_manager.py)."""

    def __init__(self, index=None, clients=None, start_thread=True):
        self.index = index
        self.channelizers = {}
        self.continue_running = True
        if clients is None:
            import redis
            clients = [redis.StrictRedis(host='127.0.0.1', port='6379', db=0)]
        self.clients = clients
        if start_thread:
            t = threading.Thread(target=self.manager_loop)
            t.daemon = True
            t.start()

    def get_instance(self, instance_id):
        return self.channelizers.get(instance_id, False)

    def get_channelizer_for_frequency(self, frequency):
        """redis_channelizer_manager.py:52-76: among channelizers with a source covering `frequency`
        (abs(center - f) < samp_rate/2), pick uniformly among those with the smallest abs offset."""
        options = {}
        smallest_offset = None
        for channelizer in self.channelizers:
            for source in self.channelizers[channelizer]['sources']:
                if abs(source[0] - frequency) < source[1] / 2:
                    if smallest_offset is None or (abs(source[0] - frequency) < smallest_offset):
                        smallest_offset = abs(source[0] - frequency)
                    options.setdefault(abs(source[0] - frequency), []).append(channelizer)
        try:
            channelizer = random.choice(options[smallest_offset])
            return (self.channelizers[channelizer]['address'], self.channelizers[channelizer]['port'])
        except Exception:
            return (None, None)

    def poll_once(self, now=None):
        now = time.time() if now is None else now
        for client in self.clients:
            try:
                instances = client.smembers('channelizers')
            except Exception:
                instances = set()
            channelizers = {}
            for instance_uuid in instances:
                try:
                    data = client.get(instance_uuid)
                    if data is None:
                        continue
                    channelizer = json.loads(data)
                    if self.index is None or int(channelizer['index']) == self.index:
                        key = instance_uuid.decode('utf-8') if isinstance(instance_uuid, bytes) else instance_uuid
                        channelizers[key] = channelizer
                except Exception:
                    pass
            for name in list(channelizers):
                if channelizers[name]['current_time'] < now - 5:          # :106-110
                    client.srem('channelizers', name)
                    client.delete(name)
                    del channelizers[name]
            self.channelizers = channelizers

    def manager_loop(self):
        while self.continue_running:
            self.poll_once()
            time.sleep(0.5)
