"""`python -m rcf.frontend -i 0` as a real process on the GPU (the reference's unit of deployment:
rc_frontend/receiver.py:477-700, systemd/radiocapture-channelizer@.service:11) and a backend-shaped client in this
one: registry -> frontend_connector.create_channel -> bytes off the data wire == the oracle's channel, heartbeat expiry,
registry expiry.  Plain-TCP transports (rcf.transport): pyzmq / redis-py are not in the image."""
import json
import os
import signal
import subprocess
import sys
import time

import numpy as np
import pytest

from oracle import cbind as OC
from oracle import grspec as G
from rcf import frontend_connector as FC, registry, sources, transport

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FS, CR, FC0, F_OFF = 2400000, 12500, 855050000, -62500          # BASELINE configs[0]: cfg1's carrier

CONFIG = '''
class rc_config:
    def __init__(self):
        self.receiver_split2 = False
        self.frontend_mode = 'xlat'
        self.sources = {
            0: {'type': 'synthetic', 'center_freq': %d, 'samp_rate': %d, 'seed': 1001, 'tile_samples': 1 << 20,
                'wire': %r, 'block_ms': 20.0,
                'carriers': [{'f_off': %d, 'f_mod': 1000.0, 'dev': 2500.0, 'snr_db': 30.0}]},
            1: {'type': 'synthetic', 'center_freq': 900000000, 'samp_rate': %d},
        }
'''


def _wait(cond, timeout, what):
    t0 = time.time()
    while True:
        v = cond()
        if v:
            return v
        assert time.time() - t0 < timeout, what
        time.sleep(0.05)


@pytest.mark.parametrize("wire", ["cf32", "u8"])
def test_channelizer_process_serves_a_backend(gpu_required, tmp_path, wire):
    cfg = tmp_path / "config.py"
    cfg.write_text(CONFIG % (FC0, FS, wire, F_OFF, FS))
    ready = tmp_path / "ready.json"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "radiocapture-rf_amd"), ROOT]))
    log = open(tmp_path / "daemon.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "rcf.frontend", "-i", "0", "--config", str(cfg), "--transport", "tcp",
                             "--registry", "dir:%s" % (tmp_path / "reg"), "--bind", "127.0.0.1", "--ready-file", str(ready)],
                            env=env, cwd=str(tmp_path), stdout=log, stderr=subprocess.STDOUT)
    try:
        _wait(lambda: ready.exists() or proc.poll() is not None, 120, "daemon did not come up")
        assert proc.poll() is None, open(tmp_path / "daemon.log").read()
        info = json.loads(ready.read_text())
        reg = transport.DirRegistryClient(str(tmp_path / "reg"))
        mgr = registry.redis_channelizer_manager(index=0, clients=[reg], start_thread=False)
        _wait(lambda: (mgr.poll_once(), mgr.channelizers)[1], 10, "no registry record")
        rec = next(iter(mgr.channelizers.values()))
        assert rec["port"] == info["port"] and rec["pid"] == info["pid"] and rec["index"] == "0"
        assert rec["sources"] == [[FC0, FS]] and rec["source_count"] == 1        # -i 0 deleted the other source
        fc = FC.frontend_connector("gpu-test", mgr, transport_factory=transport.tcp_req_factory)
        chan, port = fc.create_channel(CR, FC0 + F_OFF)
        assert chan, "create_channel failed"
        sub = transport.TcpSubSocket(fc.host, port)
        n = 20000
        got = np.frombuffer(sub.recv_exact(8 * n), dtype=np.complex64)
        sub.close()
        start, decim = _wait(lambda: (mgr.poll_once(), next(iter(mgr.channelizers.values())).get(
            "rcf_channel_starts", {}).get(chan))[1], 5, "the channel's start sample never reached the registry")
        D, taps = G.channel_params(FS, CR)
        assert decim == D == 96
        # the stream the daemon's source delivers: the config's tile, looped; in wire format where the link has one
        src = dict(type="synthetic", samp_rate=FS, seed=1001, tile_samples=1 << 20, wire=wire,
                   carriers=[dict(f_off=F_OFF, f_mod=1000.0, dev=2500.0, snr_db=30.0)])
        tile = sources.synthetic_tile(src)
        if wire != "cf32":
            scale, off = sources.WIRE_SCALE[wire]
            raw = sources.to_wire(tile, wire).astype(np.float32)
            tile = ((raw - np.float32(off)) * np.float32(scale)).view(np.complex64)   # rcf_push_raw's conversion
        need = (n + 100000) * D
        reps = (start % len(tile) + need) // len(tile) + 2
        x = np.tile(tile, reps)[start % len(tile):][:need]
        ct, incr = OC.xlating_composite(taps, D, float(F_OFF), float(FS))
        want, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[1.0])
        want = want[0]
        probe = 64
        win = np.lib.stride_tricks.sliding_window_view(want[: len(want) - n + probe], probe)
        k0 = int(np.argmin(np.abs(win - got[:probe]).sum(axis=1)))
        ref = want[k0:k0 + n]
        err = float(np.sqrt(np.mean(np.abs(got - ref) ** 2) / np.mean(np.abs(ref) ** 2)))
        assert err < 1e-5, (wire, k0, err)
        # a second subscriber joins the same stream later and sees the same samples further on
        # heartbeats: the connector's own thread has been sending them; the daemon kept the channel
        rec = (mgr.poll_once(), next(iter(mgr.channelizers.values())))[1]
        assert rec["rcf_channels_in_use"] == 1 and rec["rcf_healthy"] and rec["rcf_msps_in"] > 0.5 * FS / 1e6
        assert rec["rcf_source_late_blocks"] == 0
        # the client dies without 'release' / 'quit': 5 s later the daemon has released its channel (receiver.py:654-668)
        fc.continue_running = False                                # stops the heartbeat thread ...
        fc.my_client_id = 10 ** 6                                   # ... whose parting 'quit' names nobody
        t0 = time.time()
        _wait(lambda: (mgr.poll_once(), next(iter(mgr.channelizers.values()))["rcf_channels_in_use"] == 0)[1], 12,
              "heartbeat expiry did not release the channel")
        assert time.time() - t0 > 3.0
        proc.send_signal(signal.SIGTERM)
        proc.wait(timeout=30)
        assert proc.returncode == 0
        mgr.poll_once(now=time.time() + 6)                         # 5 s without a refresh: the manager drops the record
        assert mgr.channelizers == {}
    finally:
        if proc.poll() is None:
            proc.kill()
        log.close()
