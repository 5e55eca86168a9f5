"""One process per GPU: NUMA pinning, the native layer (or its test stub), spawning the ranks."""
import os
import sys
import time

from .common import BENCH_PY, ROOT

def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def pin_to_gpu_numa(local_rank, pci=None):
    """Keep this process's threads on the CPUs of its GPU's NUMA node (the pump threads of the paced leg and the pinned
    buffers they first-touch then sit next to the GPU's PCIe root; the cpu_baseline leg's threads share one node's memory).
    The GPU's sysfs node comes from its PCI address (`pci`: rcf_device_pci_bus_id, asked of a child process so that THIS
    process has not started the HIP runtime yet); without it, the AMD render nodes in PCI order are taken as HIP's device
    order.  Node -1 (no NUMA information) or any failure: no pinning.  -> {numa_node, cpus, pinned, ...} for the line."""
    try:
        import glob
        dev = int(os.environ.get("RCF_BENCH_DEVICE", local_rank))
        if pci is None:
            pci = device_pci_of(dev)
        sysdir = "/sys/bus/pci/devices/%s" % pci if pci else None
        how = "hipDeviceGetPCIBusId"
        if not sysdir or not os.path.exists(os.path.join(sysdir, "numa_node")):
            nodes = []
            for d in sorted(glob.glob("/sys/class/drm/renderD*/device")):
                try:
                    if open(os.path.join(d, "vendor")).read().strip() == "0x1002":
                        nodes.append(os.path.realpath(d))
                except Exception:
                    continue
            nodes.sort(key=os.path.basename)
            if dev >= len(nodes):
                return {"numa_node": None, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
            sysdir, how = nodes[dev], "render nodes in PCI order"
        node = int(open(os.path.join(sysdir, "numa_node")).read())
        if node < 0:
            return {"numa_node": None, "cpus": len(os.sched_getaffinity(0)), "pinned": False, "pci": os.path.basename(sysdir)}
        cpus = _cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read()) & os.sched_getaffinity(0)
        if not cpus:
            return {"numa_node": node, "cpus": len(os.sched_getaffinity(0)), "pinned": False}
        os.sched_setaffinity(0, cpus)
        return {"numa_node": node, "cpus": len(cpus), "pinned": True, "pci": os.path.basename(sysdir), "device_found_by": how}
    except Exception as e:
        return {"numa_node": None, "cpus": None, "pinned": False, "error": "%s: %s" % (type(e).__name__, e)}


def device_pci_of(dev, timeout_s=120):
    """PCI address of HIP device `dev`, asked of a short child process (the HIP runtime's own threads inherit the affinity
    of the thread that starts it: this process pins itself first and loads librcf after)"""
    import subprocess
    if os.environ.get("RCF_BENCH_NATIVE"):
        return None
    code = ("import sys; sys.path[:0] = %r; from rcf import native; print(native.device_pci_bus_id(%d) or '')"
            % ([ROOT, os.path.join(ROOT, "radiocapture-rf_amd")], dev))
    try:
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout_s)
        out = r.stdout.strip().splitlines()
        return out[-1] if r.returncode == 0 and out and ":" in out[-1] else None
    except Exception:
        return None


def load_native():
    """librcf's ctypes layer.  RCF_BENCH_NATIVE=<module> swaps in another module with the same surface: the CPU test of
    the launcher (tests/test_bench_launcher.py) runs the whole N-rank protocol over a stub Frontend that way."""
    name = os.environ.get("RCF_BENCH_NATIVE")
    if name:
        import importlib
        return importlib.import_module(name)
    from rcf import native
    return native


def spawn_ranks(n, argv, timeout_s=3600.0):
    """--gpus n > 1 and no launcher around us: start n rank processes of this script (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set, HIP_VISIBLE_DEVICES untouched, device = local rank), relay rank 0's stdout, return
    the first non-zero exit code (the other ranks are then terminated by pid)."""
    import socket
    import subprocess
    native = load_native()
    have = native.device_count()
    if have < n and "RCF_BENCH_DEVICE" not in os.environ:
        print("bench.py: --gpus %d but %d HIP device(s) visible: refusing to measure fewer GPUs than asked for "
              "(RCF_BENCH_DEVICE=<d> runs every rank on device d)" % (n, have), file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RCF_BENCH_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    deadline = time.time() + timeout_s
    rc, out0 = 0, b""
    pending = set(range(n))
    try:
        while pending and rc == 0:
            for r in sorted(pending):
                if r == 0:
                    try:                               # drain rank 0's pipe while waiting (its line can be > 64 KB)
                        o, _ = procs[0].communicate(timeout=0.2)
                        out0 += o or b""
                    except subprocess.TimeoutExpired:
                        continue
                code = procs[r].poll()
                if code is None:
                    continue
                pending.discard(r)
                if code != 0:
                    rc = code
                    print("bench.py: rank %d exited with %d" % (r, code), file=sys.stderr)
                    break
            if time.time() > deadline:
                rc = 124
                print("bench.py: ranks still running after %.0f s" % timeout_s, file=sys.stderr)
            time.sleep(0.05)
    finally:
        for r in pending:
            if procs[r].poll() is None:
                procs[r].terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    if rc == 0 and not out0.strip():
        print("bench.py: rank 0 printed no line", file=sys.stderr)
        rc = 1
    return rc
