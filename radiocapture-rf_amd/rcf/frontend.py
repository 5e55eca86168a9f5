"""The channelizer process: `python -m rcf.frontend -i <index>` is what `rc_frontend/receiver.py -i <index>` is to the
reference (/root/reference/rc_frontend/receiver.py:477-700; one instance per SDR source:
systemd/radiocapture-channelizer@.service:11).  It assembles the parts the same way the reference's `__main__` does:

  receiver.py:487-495   argparse `-i/--index`, logging from config.logging.json     -> main() below
  receiver.py:23,57     `from config import rc_config`; `config = rc_config()`      -> load_config()
  receiver.py:497       tb = receiver(index)                                        -> rcf.receiver.receiver(config, index, device)
  receiver.py:44-46     REP socket on tcp://0.0.0.0:0                               -> FrontendServer over ZeroMQ (serve_zmq) or TCP frames
  receiver.py:268       redis_channel_publisher(sources, channels, zmq_socket, index) -> rcf.registry.redis_channel_publisher(..., extra=metrics, health=healthy)
  receiver.py:271       tb.start(): GNU Radio's scheduler threads move the samples  -> rcf.dataplane.NativeDataPlane: one native pump thread
                                                                                       (rcf_pump_t) per class of sources, no interpreter in it
  channel.py:36         one zeromq.pub_sink per channel                             -> PUB sockets fed from the pump's per-channel host rings
  receiver.py:74-204    SDR source blocks                                           -> pinned source rings: 'synthetic' (the pump's clock replays
                                                                                       the tile) | 'file' (a feeder thread) | receiver.feed
  (`--dataplane python`: the earlier arrangement -- rcf.sources.PacedSource threads call receiver.feed, rcf.egress.EgressPump
   reads the channel rings behind a device synchronisation; what receiver_split2 configurations and stub front-ends run on)
  receiver.py:616-699   the main loop (status, idle sweep, heartbeat expiry, recv/handle/send) -> FrontendServer.tick + handle

One process drives one GPU (`--device`, default: index modulo the visible devices).  ZeroMQ and Redis are used when
pyzmq / redis-py are importable (`--transport zmq`, `--registry redis`, the defaults then); otherwise, or on request,
the plain-socket stand-ins of rcf.transport carry the same strings and bytes (`--transport tcp`, `--registry dir:<path>`).
"""
from __future__ import annotations

import argparse
import importlib
import importlib.util
import json
import logging
import logging.config
import os
import signal
import sys
import threading
import time


def load_config(spec="config"):
    """`from config import rc_config` (receiver.py:23) -- `spec` is a module name on sys.path or a path to a .py file
    (configs/*.py in the reference); returns rc_config()."""
    path = spec if (spec.endswith(".py") or os.sep in spec) else None
    if path is None:
        try:
            return importlib.import_module(spec).rc_config()
        except TabError:
            path = importlib.util.find_spec(spec).origin
    # Most of the reference's configs/*.py mix tabs and spaces the way Python 2 allowed (a tab = the next multiple of
    # eight columns) and do not import under Python 3 -- the reference itself cannot load them any more.  Read them the
    # Python-2 way instead of asking the operator to re-indent a site's configuration.
    import types
    with open(path, "rt") as fh:
        src = fh.read()
    try:
        code = compile(src, path, "exec")
    except TabError:
        code = compile(src.expandtabs(8), path, "exec")
    mod = types.ModuleType("config")
    mod.__file__ = path
    exec(code, mod.__dict__)
    return mod.rc_config()


def have(module):
    try:
        importlib.import_module(module)
        return True
    except Exception:
        return False


class Daemon:
    """Everything `receiver.py -i <index>` runs, as an object (so that tests can run it in-process too)."""

    def __init__(self, config, index=None, device=None, transport=None, registry=None, bind="0.0.0.0", port=0,
                 egress_period=0.01, fm_gain=None, block_ms=20.0, frontend_factory=None, start_sources=True,
                 kernel_metrics=32, dataplane=None, max_channels=1024, pump_cpu=-1):
        from . import egress, protocol, receiver, registry as registry_mod, sources, transport as tr
        self.log = logging.getLogger("frontend" if index is None else "frontend-%s" % index)
        transport = transport or ("zmq" if have("zmq") else "tcp")
        registry = registry or ("redis" if have("redis") else "none")
        if device is None:
            device = 0
            if frontend_factory is None:
                from . import native
                n = native.device_count()
                if n < 1:
                    raise RuntimeError("no HIP device visible: the channelizer needs an MI355X")
                device = (int(index) if index is not None else 0) % n
        self.transport, self.device, self.index = transport, device, index
        self.tb = receiver.receiver(config, index=index, frontend_factory=frontend_factory, device=device)
        make_socket = egress.zmq_pub_factory() if transport == "zmq" else tr.tcp_pub_factory()
        # who moves the samples: the native pump wherever the front-ends are native ones and the sources whole (default);
        # the Python threads for receiver_split2 halves and injected stub front-ends
        pumpable = frontend_factory is None and not getattr(config, "receiver_split2", False)
        if dataplane is None:
            dataplane = "pump" if pumpable else "python"
        if dataplane == "pump" and not pumpable:
            raise ValueError("--dataplane pump needs native front-ends and whole sources (no receiver_split2)")
        self.dataplane = dataplane
        if dataplane == "pump":
            from . import dataplane as dp
            self.pump = dp.NativeDataPlane(self.tb, socket_factory=make_socket, period=egress_period, fm_gain=fm_gain,
                                           block_ms=block_ms, max_channels=max_channels, cpu=pump_cpu)
        elif dataplane == "python":
            self.pump = egress.EgressPump(self.tb, socket_factory=make_socket, period=egress_period, fm_gain=fm_gain)
        else:
            raise ValueError("dataplane %r" % dataplane)
        self.server = protocol.FrontendServer(self.tb)
        self.last_metrics = {}
        self.server.status_extra = lambda: {k: v for k, v in self.last_metrics.items() if k != "rcf_channel_starts"}
        if kernel_metrics and frontend_factory is None:
            self.tb.enable_kernel_metrics(kernel_metrics)
        from . import hostinfo
        self.cpu_meter = hostinfo.CpuMeter()
        self.stop_flag = threading.Event()
        self.rep = None
        self._zmq_thread = None
        if transport == "zmq":
            # serve_zmq binds inside its loop; run it in its own thread and wait for the endpoint
            self._zmq_thread = threading.Thread(target=self.server.serve_zmq,
                                                kwargs=dict(bind="tcp://%s:%s" % (bind, port or 0), stop=self.stop_flag.is_set),
                                                name="rep", daemon=True)
            self._zmq_thread.start()
            t0 = time.time()
            while not hasattr(self.server, "endpoint"):
                if time.time() - t0 > 10 or not self._zmq_thread.is_alive():
                    raise RuntimeError("REP socket did not come up")
                time.sleep(0.005)
            self.port = int(self.server.endpoint.rsplit(":", 1)[1])
        elif transport == "tcp":
            self.rep = tr.TcpRepServer(bind, port)
            self.port = self.rep.port
        else:
            raise ValueError("transport %r" % transport)
        client = tr.registry_client(registry)
        self.publisher = None
        if client is not None:
            self.publisher = registry_mod.redis_channel_publisher(
                sources=self.tb.sources, channels=self.tb.channels, port=self.port, index=index, client=client,
                address="127.0.0.1" if bind in ("127.0.0.1", "localhost") else None,
                extra=self.metrics, health=self.tb.healthy)
        self.pump.start()
        self.sources = []
        if start_sources and dataplane == "python":
            self.sources = sources.start_paced_sources(self.tb, block_ms=block_ms)
        self.log.info("channelizer up: index %s device %s control port %s (%s) data plane %s sources %s" % (
            index, device, self.port, transport, dataplane,
            {k: (v["center_freq"], v["samp_rate"]) for k, v in self.tb.sources.items()}))

    def metrics(self):
        pump_stats = self.pump.stats() if self.dataplane == "pump" else {}     # (also refreshes the samples-in count)
        m = self.tb.metrics()
        m.update(pump_stats)
        m.update(self.cpu_meter.sample())            # what the host gave the process: cores used, quota, throttled time
        if pump_stats.get("rcf_pump_error") and self.tb.fault is None:
            self.tb.fault = "pump: %s" % pump_stats["rcf_pump_error"]          # healthy() -> False: the heartbeat stops
        m["rcf_egress_bytes"] = self.pump.bytes_out
        m["rcf_egress_errors"] = self.pump.errors
        if self.sources:
            m["rcf_source_late_blocks"] = sum(s.late for s in self.sources)
        elif pump_stats:
            m["rcf_source_late_blocks"] = pump_stats.get("rcf_pump_late", 0) + pump_stats.get("rcf_pump_overruns", 0)
        # the data wire has no timestamps: where each channel's stream starts, for consumers that care
        with self.tb.access_lock:
            m["rcf_channel_starts"] = {b: [c.start_sample, c.decim] for b, c in self.tb.channels.items()
                                       if getattr(c, "start_sample", None) is not None}
        self.last_metrics = m
        return m

    def serve_forever(self):
        """the reference's `while 1:` (receiver.py:620-699) until stop()"""
        from . import transport as tr
        try:
            if self.rep is not None:
                tr.serve_tcp(self.server, self.rep, stop=self.stop_flag.is_set, log=self.log)
            else:
                while not self.stop_flag.is_set() and self._zmq_thread.is_alive():
                    time.sleep(0.05)
        finally:
            self.close()

    def stop(self, *a):
        self.stop_flag.set()

    def close(self):
        self.stop_flag.set()
        for s in self.sources:
            s.stop()
        self.sources = []
        if self.publisher is not None:
            self.publisher.continue_running = False
        self.pump.stop()
        if self._zmq_thread is not None:
            self._zmq_thread.join(timeout=2)
        if self.rep is not None:
            self.rep.close()
            self.rep = None
        self.tb.close()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m rcf.frontend",
                                 description="MI355X channelizer process (drop-in for rc_frontend/receiver.py -i <index>)")
    ap.add_argument("-i", "--index", help="Device config index, if specified, all other configured sources will be deleted")
    ap.add_argument("--config", default="config", help="module name or path of the file that defines rc_config (default: config)")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: index modulo visible devices)")
    ap.add_argument("--transport", choices=["zmq", "tcp"], default=None, help="control + data wire (default: zmq when pyzmq is importable)")
    ap.add_argument("--registry", default=None, help="'redis' | 'dir:<path>' | 'none' (default: redis when redis-py is importable)")
    ap.add_argument("--bind", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=0, help="control port (default: ephemeral, advertised through the registry)")
    ap.add_argument("--block-ms", type=float, default=20.0, help="block length of the paced 'synthetic' / 'file' sources")
    ap.add_argument("--fm-gain", type=float, default=None, help="also publish quadrature_demod_cf(gain) of every channel on port + 1")
    ap.add_argument("--kernel-metrics", type=int, default=32,
                    help="time every n-th kernel launch for the status line / registry record (0 = off)")
    ap.add_argument("--dataplane", choices=["pump", "python"], default=None,
                    help="who moves the samples: the native pump thread (default) or Python source / egress threads")
    ap.add_argument("--max-channels", type=int, default=1024, help="subscription slots of the native pump (channels delivered at once)")
    ap.add_argument("--pump-cpu", type=int, default=-1, help="pin the pump thread to this CPU (default: leave it to the scheduler)")
    ap.add_argument("--ready-file", default=None, help="write {'port':..,'pid':..} here once the control port is bound")
    args = ap.parse_args(argv)

    configured = False
    if os.path.exists("config.logging.json"):                   # receiver.py:478-487
        try:
            with open("config.logging.json", "rt") as f:
                logging.config.dictConfig(json.load(f))
            configured = True
        except Exception as e:                                   # e.g. its rotating-file handler's ../logs/ is not there:
            sys.stderr.write("config.logging.json not usable (%s): logging to stderr\n" % e)   # the reference dies here
    if not configured:
        logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(levelname)s %(message)s")

    config = load_config(args.config)
    d = Daemon(config, index=args.index, device=args.device, transport=args.transport, registry=args.registry,
               bind=args.bind, port=args.port, fm_gain=args.fm_gain, block_ms=args.block_ms, kernel_metrics=args.kernel_metrics,
               dataplane=args.dataplane, max_channels=args.max_channels, pump_cpu=args.pump_cpu)
    signal.signal(signal.SIGTERM, d.stop)
    signal.signal(signal.SIGINT, d.stop)
    if args.ready_file:
        tmp = args.ready_file + ".tmp"
        with open(tmp, "w") as f:
            json.dump({"port": d.port, "pid": os.getpid(), "transport": d.transport, "device": d.device}, f)
        os.replace(tmp, args.ready_file)
    d.serve_forever()
    return 0


if __name__ == "__main__":
    sys.exit(main())
