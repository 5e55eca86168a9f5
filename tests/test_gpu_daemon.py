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


def _host_view(rec):
    return {k: rec.get(k) for k in rec if k.startswith("rcf_host_") or k.startswith("rcf_pump_late_wakeups")}


def _late_is_the_hosts(rec, rec0=None):
    """a late block is the library's only when the pump thread had a CPU: the record carries how long the pump's late
    wake-ups were and how much of that time the thread sat on a run queue, and the cgroup's throttled time (this container
    has a CFS quota, neighbours, and no SCHED_FIFO)"""
    def d(k):
        return (rec.get(k) or 0.0) - ((rec0 or {}).get(k) or 0.0)
    slow, on_rq = d("rcf_pump_late_wakeups_ms"), d("rcf_pump_late_wakeups_on_run_queue_ms")
    return (slow > 0 and on_rq > 0.9 * slow) or d("rcf_host_throttled_ms") > 0


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
        # (the channel's outputs sit on the absolute decimation grid: output k ends at sample k D of the source's stream, the
        # first one at the first multiple of D at or after the channel's start)
        start = -(-start // D) * D
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
        # (on the native pump a subscriber that connects at once sees the channel's very first outputs: their filter
        # history reaches before the channel's start, where the device has zeros and this oracle run -- started on the
        # decimation grid, up to D - 1 samples later -- has not: the first ntaps / D outputs are left out)
        lead = len(taps) // D + 2
        err = float(np.sqrt(np.mean(np.abs(got[lead:] - ref[lead:]) ** 2) / np.mean(np.abs(ref[lead:]) ** 2)))
        assert err < 1e-5, (wire, k0, err)
        # a second subscriber joins the same stream later and sees the same samples further on
        # heartbeats: the connector's own thread has been sending them; the daemon kept the channel
        rec = (mgr.poll_once(), next(iter(mgr.channelizers.values())))[1]
        assert rec["rcf_channels_in_use"] == 1 and rec["rcf_healthy"] and rec["rcf_msps_in"] > 0.5 * FS / 1e6
        # (a block is late when its delivery STARTS more than a block period after its last sample exists; a source that did
        # not keep up would be late on every block from then on -- a handful of late ones is the host's scheduler, and this
        # test shares its cores with whatever else the suite runs)
        assert rec["rcf_source_late_blocks"] <= 5 or _late_is_the_hosts(rec), (rec["rcf_source_late_blocks"], _host_view(rec))
        # kernel time in the record -- every 32nd launch of each kernel class is timed
        assert rec["rcf_kernel_us"].get("fir", 0) > 0 and 0 < rec["rcf_gpu_busy_fraction_est"] < 1
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


CONFIG_PFB = '''
class rc_config:
    def __init__(self):
        self.receiver_split2 = False
        self.frontend_mode = 'pfb'
        self.sources = {
            0: {'type': 'file', 'path': %r, 'format': 's16', 'loop': True, 'center_freq': %d, 'samp_rate': %d, 'block_ms': 10.0},
        }
'''


def test_channelizer_process_in_pfb_mode_replaying_a_capture_file(gpu_required, tmp_path):
    """the same process with `frontend_mode = 'pfb'` (the reference's unfinished branch, receiver.py:242-261,343-383)
    and a `'file'` source in the s16 wire format: an on-grid request is served by a bin of the 400-bin bank (the registry
    metrics say so), an off-grid one by the direct kernel, and both streams off the data wire equal the GR-faithful
    oracle's channels over the looped capture."""
    fs, fc = 5000000, 460000000
    D, taps = G.channel_params(fs, CR)                                   # 200, 727: the bank is 400 bins
    rng = np.random.default_rng(77)
    n = 1 << 20
    from rcf import synth
    x = synth.awgn(rng, n).astype(np.complex128)
    for f_off in (25 * 12500.0, -31 * 12500.0 + 300.0):
        x += synth.nbfm_carrier(n, fs, round(f_off * n / fs) * fs / n, round(900.0 * n / fs) * fs / n, 2500.0,
                                synth.snr_amp(30.0, 12500.0, fs))
    raw = sources.to_wire(x.astype(np.complex64), "s16")
    cap = tmp_path / "capture.s16"
    raw.tofile(cap)
    scale, off = sources.WIRE_SCALE["s16"]
    tile = ((raw.astype(np.float32) - np.float32(off)) * np.float32(scale)).view(np.complex64)
    cfg = tmp_path / "config.py"
    cfg.write_text(CONFIG_PFB % (str(cap), fc, fs))
    ready = tmp_path / "ready.json"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "radiocapture-rf_amd"), ROOT]))
    log = open(tmp_path / "daemon.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "rcf.frontend", "-i", "0", "--config", str(cfg), "--transport", "tcp",
                             "--registry", "dir:%s" % (tmp_path / "reg"), "--bind", "127.0.0.1", "--ready-file", str(ready)],
                            env=env, cwd=str(tmp_path), stdout=log, stderr=subprocess.STDOUT)
    try:
        _wait(lambda: ready.exists() or proc.poll() is not None, 120, "daemon did not come up")
        assert proc.poll() is None, open(tmp_path / "daemon.log").read()
        reg = transport.DirRegistryClient(str(tmp_path / "reg"))
        mgr = registry.redis_channelizer_manager(index=0, clients=[reg], start_thread=False)
        _wait(lambda: (mgr.poll_once(), mgr.channelizers)[1], 10, "no registry record")
        clients, got, chans = [], {}, {}
        n_out = 12000
        for name, f_off in (("bank", 25 * 12500), ("direct", -31 * 12500 + 300)):
            fc_ = FC.frontend_connector("gpu-test-%s" % name, mgr, transport_factory=transport.tcp_req_factory)
            chan, port = fc_.create_channel(CR, fc + f_off)
            assert chan, name
            sub = transport.TcpSubSocket(fc_.host, port)
            got[name] = np.frombuffer(sub.recv_exact(8 * n_out), dtype=np.complex64)
            sub.close()
            clients.append(fc_)
            chans[name] = (chan, f_off)
        def record_with_starts():
            mgr.poll_once()
            r = next(iter(mgr.channelizers.values()))
            return r if all(c in r.get("rcf_channel_starts", {}) for c, _ in chans.values()) else None
        rec = _wait(record_with_starts, 6, "channel starts never reached the registry")
        assert rec["rcf_pfb_served_by_bank"] == 1 and rec["rcf_pfb_direct_off_grid"] == 1 and rec["rcf_pfb_direct_parity_budget"] == 0
        for name, (chan, f_off) in chans.items():
            start, decim = rec["rcf_channel_starts"][chan]
            # a bank tap's start counts the bank's FRAMES (decim 200 samples each); the direct channel's, samples
            s0 = start * D if name == "bank" else -(-start // D) * D      # (a direct channel's outputs: the absolute decimation grid)
            need = (n_out + 60000) * D
            reps = (s0 % n + need) // n + 2
            if name == "bank":
                # a bin comes with the filter's history in it (the bank was running before the tap was opened): give the
                # oracle the samples before the tap's first frame too, and drop the outputs they produce
                lead = (len(taps) // D + 2) * D
                xs = np.tile(tile, reps + 1)[n + s0 % n - lead:][:need + lead]
                skip = lead // D
            else:
                xs, skip = np.tile(tile, reps)[s0 % n:][:need], 0
            ct, incr = OC.xlating_composite(taps, D, float(f_off), float(fs))
            want, _ = OC.channel_bank(xs, D, ct[None, :], np.array([incr]), gains=[1.0])
            want = want[0][skip:]
            probe = 64
            win = np.lib.stride_tricks.sliding_window_view(np.abs(want[: len(want) - n_out + probe]), probe)
            k0 = int(np.argmin(np.abs(win - np.abs(got[name][:probe])).sum(axis=1)))
            ref = want[k0:k0 + n_out]
            # the bank tap's rotator starts at the tap's opening (GNU Radio's would have started with the flowgraph):
            # a constant phase between the two streams; remove it, then compare
            rot = np.vdot(ref, got[name]) / np.vdot(ref, ref)
            err = float(np.sqrt(np.mean(np.abs(got[name] - rot * ref) ** 2) / np.mean(np.abs(ref) ** 2)))
            assert abs(abs(rot) - 1) < 1e-4 and err < (1e-4 if name == "bank" else 1e-5), (name, k0, abs(rot), err)
        for c in clients:
            c.release_channel()
            c.exit()
        proc.send_signal(signal.SIGTERM)
        proc.wait(timeout=30)
        assert proc.returncode == 0
    finally:
        if proc.poll() is None:
            proc.kill()
        log.close()


CONFIG_20M = '''
class rc_config:
    def __init__(self):
        self.receiver_split2 = False
        self.frontend_mode = 'xlat'
        self.sources = {0: {'type': 'synthetic', 'center_freq': 860000000, 'samp_rate': 20000000, 'seed': 5, 'tile_samples': 1 << 22,
                            'wire': 'u8', 'block_ms': 20.0, 'carriers': []}}
'''


def test_channelizer_process_keeps_up_with_one_20_msps_source_and_64_subscribed_channels(gpu_required, tmp_path):
    """the deployment shape at the BASELINE rate: ONE channelizer process, one 20 Msps source paced at wall-clock rate
    (u8 wire, 20 ms blocks), 64 reference-shaped 12.5 kHz channels (2909-tap xlating FIR /800 each: the matrix-core
    bank) created by 64 clients and every one of them subscribed to -- for eight seconds: no late source block, every
    subscriber receives its 25 kS/s, the registry says so."""
    cfg = tmp_path / "config.py"
    cfg.write_text(CONFIG_20M)
    ready = tmp_path / "ready.json"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "radiocapture-rf_amd"), ROOT]))
    log = open(tmp_path / "daemon.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "rcf.frontend", "-i", "0", "--config", str(cfg), "--transport", "tcp",
                             "--registry", "dir:%s" % (tmp_path / "reg"), "--bind", "127.0.0.1", "--ready-file", str(ready)],
                            env=env, cwd=str(tmp_path), stdout=log, stderr=subprocess.STDOUT)
    import socket
    import threading
    try:
        _wait(lambda: ready.exists() or proc.poll() is not None, 120, "daemon did not come up")
        assert proc.poll() is None, open(tmp_path / "daemon.log").read()
        reg = transport.DirRegistryClient(str(tmp_path / "reg"))
        mgr = registry.redis_channelizer_manager(index=0, clients=[reg], start_thread=False)
        _wait(lambda: (mgr.poll_once(), mgr.channelizers)[1], 10, "no registry record")
        clients, subs = [], []
        for i in range(64):
            fc_ = FC.frontend_connector("load-%d" % i, mgr, transport_factory=transport.tcp_req_factory)
            chan, port = fc_.create_channel(CR, 860000000 + (i - 32) * 250000 + 12500)
            assert chan, i
            subs.append(transport.TcpSubSocket(fc_.host, port))
            clients.append(fc_)
        got = [0] * 64
        stop = threading.Event()

        def reader(lo, hi):
            for s in subs[lo:hi]:
                s.sock.settimeout(0.05)
            while not stop.is_set():
                for i in range(lo, hi):
                    try:
                        got[i] += len(subs[i].sock.recv(1 << 16))
                    except (socket.timeout, BlockingIOError):
                        pass
        ths = [threading.Thread(target=reader, args=(k, k + 16)) for k in range(0, 64, 16)]
        for t_ in ths:
            t_.start()
        time.sleep(1.5)                                           # everybody connected, rings drained
        base = list(got)
        mgr.poll_once()
        r0 = next(iter(mgr.channelizers.values()))
        t0 = time.time()
        time.sleep(8.0)
        now = list(got)
        wall = time.time() - t0
        mgr.poll_once()
        r1 = next(iter(mgr.channelizers.values()))
        stop.set()
        for t_ in ths:
            t_.join()
        assert r1["rcf_channels_in_use"] == 64 and r1["rcf_healthy"]
        # 400 blocks in the window: a source that does not keep up is late on all of them; one per cent is scheduler noise
        assert r1["rcf_source_late_blocks"] - r0["rcf_source_late_blocks"] <= 4 or _late_is_the_hosts(r1, r0), (
            r0["rcf_source_late_blocks"], r1["rcf_source_late_blocks"], _host_view(r1))
        assert abs(r1["rcf_msps_in"] - 20.0) < 0.5, r1["rcf_msps_in"]
        rates = [(b - a) / 8.0 / wall for a, b in zip(base, now)]            # cf32 samples per second per subscriber
        tol = 0.10 if _late_is_the_hosts(r1, r0) else 0.03          # (a host that starved the pump starved the readers too)
        assert min(rates) > (1 - tol) * 25000 and max(rates) < (1 + tol) * 25000, (min(rates), max(rates), _host_view(r1))
        assert r1["rcf_egress_errors"] == 0
        for s in subs:
            s.close()
        for c in clients:
            c.exit()
        proc.send_signal(signal.SIGTERM)
        proc.wait(timeout=30)
    finally:
        if proc.poll() is None:
            proc.kill()
        log.close()


# ---------------------------------------------------------------------------------------------------------------------
# The product process on the native pump: what the reference's channelizer holds in ONE top block
# (rc_frontend/receiver.py:67-70,170-204) moved by one native thread, no interpreter between a source's ring and a
# channel's host ring.

CONFIG_TEN = '''
class rc_config:
    def __init__(self):
        self.receiver_split2 = False
        self.frontend_mode = 'xlat'
        self.sources = {}
        for i in range(10):
            self.sources[i] = {'type': 'synthetic', 'center_freq': 851000000 + 3000000 * i, 'samp_rate': 2400000,
                               'seed': 2000 + i, 'tile_samples': 1 << 20, 'wire': 'u8' if i % 2 else 'cf32', 'block_ms': 20.0,
                               'carriers': [{'f_off': 12500.0 * (3 * i - 14), 'f_mod': 700.0 + 50 * i, 'dev': 2500.0, 'snr_db': 30.0}]}
'''


def _tile_of(src):
    tile = sources.synthetic_tile(src)
    if src.get("wire", "cf32") != "cf32":
        scale, off = sources.WIRE_SCALE[src["wire"]]
        raw = sources.to_wire(tile, src["wire"]).astype(np.float32)
        tile = ((raw - np.float32(off)) * np.float32(scale)).view(np.complex64)   # rcf_push_raw's conversion
    return tile


def _stream_error(got, tile, start, f_off, fs):
    n = len(got)
    D, taps = G.channel_params(fs, CR)
    start = -(-start // D) * D                         # (outputs sit on the absolute decimation grid)
    need = (n + 100000) * D
    reps = (start % len(tile) + need) // len(tile) + 2
    x = np.tile(tile, reps)[start % len(tile):][:need]
    ct, incr = OC.xlating_composite(taps, D, float(f_off), float(fs))
    want, _ = OC.channel_bank(x, D, ct[None, :], np.array([incr]), gains=[1.0])
    want = want[0]
    probe = 64
    win = np.lib.stride_tricks.sliding_window_view(want[: len(want) - n + probe], probe)
    k0 = int(np.argmin(np.abs(win - got[:probe]).sum(axis=1)))
    ref = want[k0:k0 + n]
    lead = len(taps) // D + 2                          # (outputs whose filter history reaches before the channel's start)
    return float(np.sqrt(np.mean(np.abs(got[lead:] - ref[lead:]) ** 2) / np.mean(np.abs(ref[lead:]) ** 2))), k0


def test_channelizer_process_with_ten_sources_on_the_native_pump(gpu_required, tmp_path):
    """the reference's shipped shape (configs/config_denver_dev_den817.py:25-118: ten 2.4 Msps RTL-SDRs behind one
    channelizer host) as ONE process without -i: ten sources in two wire formats (two pump classes), every source's
    control channel requested over the control wire and subscribed to; the bytes off each data socket are the oracle's
    channel of that source's stream, the status record carries the pump's own statistics, nothing was late."""
    fs = 2400000
    cfg = tmp_path / "config.py"
    cfg.write_text(CONFIG_TEN)
    ready = tmp_path / "ready.json"
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "radiocapture-rf_amd"), ROOT]))
    log = open(tmp_path / "daemon.log", "w")
    proc = subprocess.Popen([sys.executable, "-m", "rcf.frontend", "--config", str(cfg), "--transport", "tcp",
                             "--registry", "dir:%s" % (tmp_path / "reg"), "--bind", "127.0.0.1", "--ready-file", str(ready)],
                            env=env, cwd=str(tmp_path), stdout=log, stderr=subprocess.STDOUT)
    try:
        _wait(lambda: ready.exists() or proc.poll() is not None, 180, "daemon did not come up")
        assert proc.poll() is None, open(tmp_path / "daemon.log").read()
        reg = transport.DirRegistryClient(str(tmp_path / "reg"))
        mgr = registry.redis_channelizer_manager(clients=[reg], start_thread=False)
        _wait(lambda: (mgr.poll_once(), mgr.channelizers)[1], 10, "no registry record")
        rec = next(iter(mgr.channelizers.values()))
        assert rec["source_count"] == 10
        clients, subs, chans = [], [], []
        for i in range(10):
            f_off = 12500 * (3 * i - 14)                           # (the control wire carries integers: receiver.py:520)
            fc_ = FC.frontend_connector("ten-%d" % i, mgr, transport_factory=transport.tcp_req_factory)
            chan, port = fc_.create_channel(CR, 851000000 + 3000000 * i + f_off)
            assert chan, (i, open(tmp_path / "daemon.log").read()[-3000:])
            subs.append(transport.TcpSubSocket(fc_.host, port))
            clients.append(fc_)
            chans.append((chan, f_off))
        n = 16000
        got = [np.frombuffer(s_.recv_exact(8 * n), dtype=np.complex64) for s_ in subs]

        def record_with_starts():
            mgr.poll_once()
            r = next(iter(mgr.channelizers.values()))
            return r if all(c in r.get("rcf_channel_starts", {}) for c, _ in chans) else None
        mgr.poll_once()
        rec0 = next(iter(mgr.channelizers.values()))
        time.sleep(3.0)
        mgr.poll_once()
        rec = next(iter(mgr.channelizers.values()))
        assert rec["rcf_channels_in_use"] == 10 and rec["rcf_healthy"]
        assert rec["rcf_pump_classes"] == 2 and rec["rcf_pump_subscriptions"] == 10
        assert rec["rcf_pump_blocks_done"] > 10 * 20 and rec["rcf_pump_group_blocks"] > 0
        assert "rcf_pump_error" not in rec
        # steady state (ten channels open, ten subscribers reading): ~2200 blocks of 13.7 ms in the window, none late
        late = rec["rcf_pump_late"] + rec["rcf_pump_overruns"] - rec0["rcf_pump_late"] - rec0["rcf_pump_overruns"]
        assert abs(rec["rcf_msps_in"] - 24.0) < 1.5, rec["rcf_msps_in"]
        # a late block is the library's only when the pump thread had a CPU: the record says how long its late wake-ups
        # were and how much of that the thread spent on a run queue (this container has a CFS quota and no CPUs of its own)
        on_rq = rec["rcf_pump_late_wakeups_on_run_queue_ms"] - rec0["rcf_pump_late_wakeups_on_run_queue_ms"]
        slow = rec["rcf_pump_late_wakeups_ms"] - rec0["rcf_pump_late_wakeups_ms"]
        host = {k: (rec0.get(k), rec.get(k)) for k in rec if k.startswith("rcf_host_")}
        print("ten sources: %d late blocks in 3 s; late wake-ups %.1f ms, %.1f ms of them on a run queue; host %s" % (late, slow, on_rq, host))
        assert late <= 2 or on_rq > 0.9 * slow > 0, (late, slow, on_rq, host)
        # (the oracle runs of this process come AFTER the steady-state window: they take every core of the container's quota)
        rec = _wait(record_with_starts, 6, "channel starts never reached the registry")
        for i, (chan, f_off) in enumerate(chans):
            src = dict(type="synthetic", samp_rate=fs, seed=2000 + i, tile_samples=1 << 20, wire="u8" if i % 2 else "cf32",
                       carriers=[dict(f_off=float(f_off), f_mod=700.0 + 50 * i, dev=2500.0, snr_db=30.0)])
            start, decim = rec["rcf_channel_starts"][chan]
            err, k0 = _stream_error(got[i], _tile_of(src), start, f_off, fs)
            assert err < 1e-5, (i, k0, err)
        for s_ in subs:
            s_.close()
        for c in clients:
            c.release_channel()
            c.exit()
        proc.send_signal(signal.SIGTERM)
        proc.wait(timeout=30)
        assert proc.returncode == 0
    finally:
        if proc.poll() is None:
            proc.kill()
        log.close()


class _CountingSocket:
    """a PUB socket nobody listens to: counts what the egress thread hands it"""
    made = []

    def __init__(self, port):
        self.port, self.n = port, 0
        _CountingSocket.made.append(self)

    def send(self, payload):
        self.n += len(payload)

    def close(self):
        pass


def test_data_plane_32_front_ends_at_20_msps_with_256_subscribed_channels_each(gpu_required):
    """32 x 20 Msps u8 sources in one receiver, 256 reference-shaped 12.5 kHz channels on each (8192 channel flowgraphs in
    the reference: rc_frontend/channel.py:29-38 each), all of them subscribed -- ten seconds on the native pump: no block
    late, every channel delivers its 25 kS/s to its socket."""
    from rcf import dataplane, receiver as receiver_mod

    class Cfg:
        receiver_split2 = False
        frontend_mode = "xlat"
        sources = {i: {"type": "synthetic", "center_freq": 400000000 + 25000000 * i, "samp_rate": 20000000, "seed": 50 + i,
                       "tile_samples": 1 << 21, "wire": "u8", "block_ms": 20.0, "carriers": []} for i in range(32)}

    _CountingSocket.made = []
    tb = receiver_mod.receiver(Cfg(), device=0)
    plane = dataplane.NativeDataPlane(tb, socket_factory=_CountingSocket, period=0.02, max_channels=8192, out_ring_samples=1 << 13)
    try:
        for i in range(32):
            for k in range(256):
                tb.connect_channel(CR, 400000000 + 25000000 * i + (k - 128) * 62500 + 12500)
        assert len(tb.channels) == 8192
        cl, = plane.classes.values()
        period = cl.block / cl.fs
        plane.start()
        t_sub = time.time()
        _wait(lambda: plane.stats()["rcf_pump_subscriptions"] == 8192, 60, "the channels were never all subscribed")
        t_sub = time.time() - t_sub
        time.sleep(1.0)
        from rcf import hostinfo
        h0 = hostinfo.cgroup_cpu()
        s0, b0, t0 = plane.stats(), [k.n for k in _CountingSocket.made], time.time()
        time.sleep(10.0)
        s1, b1, wall = plane.stats(), [k.n for k in _CountingSocket.made], time.time() - t0
        h1 = hostinfo.cgroup_cpu()
        assert "rcf_pump_error" not in s1, s1
        blocks = s1["rcf_pump_blocks_done"] - s0["rcf_pump_blocks_done"]
        assert blocks > 0.97 * 32 * wall / period, (blocks, wall, period)
        # no block late -- unless the pump thread was runnable WITHOUT a CPU meanwhile (this container has a CFS quota and
        # neighbours: the pump's own account of its late wake-ups says how much of them it spent on a run queue)
        late = (s1["rcf_pump_late"] - s0["rcf_pump_late"]) + (s1["rcf_pump_overruns"] - s0["rcf_pump_overruns"])
        slow = s1["rcf_pump_late_wakeups_ms"] - s0["rcf_pump_late_wakeups_ms"]
        on_rq = s1["rcf_pump_late_wakeups_on_run_queue_ms"] - s0["rcf_pump_late_wakeups_on_run_queue_ms"]
        throttled = h1.get("throttled_ms", 0.0) - h0.get("throttled_ms", 0.0)
        assert late == 0 or on_rq > 0.9 * slow > 0 or throttled > 0, (late, slow, on_rq, throttled, s0, s1)
        # (random ports collide now and then: a socket bound for a port that another channel then took stays unused)
        rates = [(b - a) / 8.0 / wall for a, b in zip(b0, b1) if b > 0]
        assert len(rates) == 8192 and max(rates) < 1.05 * 25000, (len(rates), max(rates))
        assert min(rates) > (0.95 if late == 0 else 0.8) * 25000, (min(rates), late)
        assert plane.errors == 0 and tb.healthy()
        print("32 x 20 Msps, 8192 channels: %d blocks of %.1f ms in %.1f s, latency p99 %.2f ms max %.2f ms, subscribing took %.1f s" % (
            blocks, period * 1e3, wall, s1["rcf_pump_latency_ms_p99"], s1["rcf_pump_latency_ms_max"], t_sub))
    finally:
        plane.stop()
        tb.close()


class _CollectingSocket:
    made = []

    def __init__(self, port):
        self.port, self.chunks = port, []
        _CollectingSocket.made.append(self)

    def send(self, payload):
        self.chunks.append(payload)

    def close(self):
        pass


def test_data_plane_takes_receiver_feed_from_whatever_owns_the_sdr(gpu_required):
    """a source whose 'type' has no driver here (the reference's rtlsdr / usrp / bladerf blocks, receiver.py:74-191): the
    owner of the device hands samples to receiver.feed / feed_raw as they arrive -- in pieces of any size -- and under the
    native data plane they land in the source's counter-fed ring, which the pump takes block by block.  Two sources in two
    wire formats; what leaves the channels' sockets is the oracle's channel of the fed stream."""
    import threading
    from rcf import dataplane, receiver as receiver_mod
    fs = 2400000
    D, taps = G.channel_params(fs, CR)

    class Cfg:
        receiver_split2 = False
        frontend_mode = "xlat"
        sources = {0: {"type": "usrp", "center_freq": 855000000, "samp_rate": fs, "block_ms": 20.0},
                   1: {"type": "rtlsdr", "center_freq": 860000000, "samp_rate": fs, "block_ms": 20.0, "wire": "u8"}}

    offs = [-62500, 137500]
    streams = []
    for i in range(2):
        src = dict(type="synthetic", samp_rate=fs, seed=3100 + i, tile_samples=1 << 20,
                   carriers=[dict(f_off=float(offs[i]), f_mod=800.0 + 100 * i, dev=2500.0, snr_db=30.0)])
        streams.append(sources.synthetic_tile(src))
    raw1 = sources.to_wire(streams[1], "u8")
    scale, off = sources.WIRE_SCALE["u8"]
    x1 = ((raw1.astype(np.float32) - np.float32(off)) * np.float32(scale)).view(np.complex64)
    _CollectingSocket.made = []
    tb = receiver_mod.receiver(Cfg(), device=0)
    plane = dataplane.NativeDataPlane(tb, socket_factory=_CollectingSocket, period=0.01)
    assert len(plane.classes) == 2
    ids = [tb.connect_channel(CR, 855000000 + offs[0])[0], tb.connect_channel(CR, 860000000 + offs[1])[0]]
    starts = [tb.channels[b].start_sample for b in ids]
    assert starts == [0, 0]
    plane.start()
    n_blocks_fed = 20
    n_total = n_blocks_fed * 48000                                 # 0.4 s of signal per source (inside the 2^20-sample streams)
    stop = threading.Event()

    failed = []

    def feeder(i):
        try:
            feed_all(i)
        except Exception as e:                                     # (a thread's exception would otherwise go unseen)
            failed.append((i, repr(e)))

    def feed_all(i):
        # pieces of 10 007 samples (nothing to do with the pump's 48 000-sample blocks), at the source's rate
        piece, at, t0 = 10007, 0, time.perf_counter()
        while at < n_total and not stop.is_set():
            n = min(piece, n_total - at)
            due = t0 + (at + n) / fs
            now = time.perf_counter()
            if now < due:
                time.sleep(due - now)
            if i == 0:
                tb.feed(0, streams[0][at:at + n])
            else:
                tb.feed_raw(1, raw1[2 * at:2 * (at + n)], 1, scale, off)
            at += n
    ths = [threading.Thread(target=feeder, args=(i,)) for i in range(2)]
    try:
        with pytest.raises(ValueError):
            tb.feed_raw(0, raw1[:100], 1, scale, off)              # source 0 is configured for cf32
        for t_ in ths:
            t_.start()
        for t_ in ths:
            t_.join(timeout=30)
        _wait(lambda: plane.stats()["rcf_pump_blocks_done"] >= 2 * n_blocks_fed, 20, "the pump never took all the blocks that were fed")
        time.sleep(0.2)                                            # (the egress thread's last pass)
        st = plane.stats()
        assert not failed, failed
        assert "rcf_pump_error" not in st and st["rcf_pump_blocks_done"] == 2 * n_blocks_fed, (st, plane.detail())
    finally:
        stop.set()
        plane.stop()
        tb.close()
    # (sockets are made in channel order: the first made belongs to the first channel)
    used = [s_ for s_ in _CollectingSocket.made if s_.chunks]
    assert len(used) == 2
    for i, x in enumerate((streams[0], x1)):
        got = np.frombuffer(b"".join(used[i].chunks), dtype=np.complex64)
        ct, incr = OC.xlating_composite(taps, D, float(offs[i]), float(fs))
        want, _ = OC.channel_bank(x[:n_total], D, ct[None, :], np.array([incr]), gains=[1.0])
        want = want[0]
        assert len(got) == len(want) == n_total // D, (i, len(got), len(want))
        err = float(np.sqrt(np.mean(np.abs(got - want) ** 2) / np.mean(np.abs(want) ** 2)))
        assert err < 1e-5, (i, err)
