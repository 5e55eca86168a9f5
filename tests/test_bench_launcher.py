"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (the reference
starts one channelizer process per source: rc_frontend/receiver.py:67-70).  Run here end to end on a stub of
librcf's Python layer (tests/stub_native.py): spawn, host rendezvous, max-over-ranks timing, peak gather, ONE line."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--steps", "6", "--warmup", "1", "--block", str(1 << 16), "--prewarm-seconds", "0", "--no-extras",
         "--no-cpu-baseline", "--no-sustained"]


def _run(gpus, extra_env=None, timeout=180):
    env = dict(os.environ, RCF_BENCH_NATIVE="stub_native", RCF_BENCH_TRANSPORT="host",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + FLAGS,
                          env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_invocation_with_gpus_2_starts_two_ranks_and_prints_one_line():
    r = _run(2)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 6
    assert d["transport"] == "host-tcp" and d["ranks_started_by"].startswith("bench.py itself")
    assert len(d["ms_per_step_by_rank"]) == 2
    # the stub's rank 1 sleeps twice as long per commit: the line's time is the MAX over ranks, and value the whole job
    assert d["ms_per_step"] >= max(d["ms_per_step_by_rank"]) * 0.999
    assert d["ms_per_step_by_rank"][1] > d["ms_per_step_by_rank"][0]
    total = 2 * 6 * (1 << 16)
    assert abs(d["value"] - total / (d["ms_per_step"] * 6e-3) / 1e6) / d["value"] < 1e-6
    assert d["peaks_allgather"]["ranks"] == 2 and d["peaks_allgather"]["values_gathered"] >= 2


def test_fewer_devices_than_ranks_is_refused_not_downgraded():
    r = _run(2, {"STUB_DEVICES": "1"})
    assert r.returncode == 2 and not r.stdout.strip()
    assert "refusing" in r.stderr
    r = _run(2, {"STUB_DEVICES": "1", "RCF_BENCH_DEVICE": "0"})       # every rank on device 0, on request
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout)["n_gpus"] == 2


def test_a_failing_rank_fails_the_whole_run():
    r = _run(2, {"STUB_FAIL_RANK": "1"}, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()


def test_one_gpu_line_is_unchanged_by_the_launcher():
    r = _run(1)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 1 and d["transport"].startswith("none") and d["rccl_ranks"] == 1


def test_the_drivers_torchrun_invocation_for_n_gt_1():
    """the contract's other launch form: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...` -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher,
    bench.py must NOT spawn again, rank 0 prints the one line"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RCF_BENCH_NATIVE="stub_native", RCF_BENCH_TRANSPORT="host",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2"] + FLAGS, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_started_by"].startswith("the launcher") and len(d["ms_per_step_by_rank"]) == 2


def test_eight_ranks_through_the_communicator_branch():
    """N = 8, the way the driver's scaling run starts it, with the stub's pretend communicator: rank 0 draws the id, the
    host rendezvous carries it, every rank joins, the all-gather + all-reduce proof runs BEFORE the warm-up, and the
    barrier / peak gather go through the communicator (bench.py's `use_rccl` branch, which a 1-GPU box cannot reach)."""
    r = _run(8, {"STUB_RCCL": "1", "STUB_DEVICES": "8", "RCF_BENCH_TRANSPORT": "rccl"}, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 8 and d["transport"] == "rccl" and d["rccl_ranks"] == 8
    assert d["rccl_proof"]["allgather_of_rank_numbers"] == list(range(8))
    assert d["rccl_proof"]["allreduce_max_of_rank_numbers"] == 7.0
    assert len(d["ms_per_step_by_rank"]) == 8 and d["ms_per_step"] >= max(d["ms_per_step_by_rank"]) * 0.999
    assert abs(d["value"] - 8 * 6 * (1 << 16) / (d["ms_per_step"] * 6e-3) / 1e6) / d["value"] < 1e-6
    assert d["peaks_allgather"]["ranks"] == 8 and d["peaks_allgather"]["values_gathered"] >= 8


def test_ranks_that_cannot_join_fall_back_to_the_host_transport_together():
    """the id is drawn and broadcast, but ncclCommInitRank fails (on every rank: a box whose RCCL cannot start): the
    ranks agree on the host transport and the line says so"""
    r = _run(4, {"STUB_RCCL": "1", "STUB_RCCL_INIT_FAILS": "1", "STUB_DEVICES": "4", "RCF_BENCH_TRANSPORT": "rccl"}, timeout=300)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 4 and d["transport"] == "host-tcp" and d["rccl_proof"] is None
    assert "could not join" in r.stderr


def test_a_communicator_that_never_comes_up_ends_the_run_instead_of_hanging_it():
    r = _run(2, {"STUB_RCCL": "1", "STUB_RCCL_HANGS": "1", "RCF_BENCH_TRANSPORT": "rccl", "RCF_BENCH_RCCL_TIMEOUT": "2"}, timeout=120)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "giving up" in r.stderr


def test_eight_rank_line_carries_every_rank_and_a_real_time_point_per_gpu():
    """VERDICT r04 item 5, the pre-flight of the first 8-GPU run: per-rank roofline fractions and sustained windows (not only
    the slowest rank's), each rank's NUMA pinning, `--config cfg5` with the scan time over ranks, the gather time and the
    gathered count checked against the sum of the ranks' peak counts, and one paced real-time point PER GPU so that the
    line answers "channels sustained" at N = 8 too -- here on the stub, one rank told to miss its deadlines"""
    flags = ["--steps", "6", "--warmup", "1", "--block", str(1 << 20), "--prewarm-seconds", "0", "--no-cpu-baseline",
             "--sustained-seconds", "0.05", "--config", "cfg5", "--rt-seconds", "0.5", "--rt-k-per-gpu", "6", "--rt-pumps", "2"]
    env = dict(os.environ, RCF_BENCH_NATIVE="stub_native", RCF_BENCH_TRANSPORT="rccl", STUB_RCCL="1", STUB_DEVICES="8", STUB_RT_MISS="5",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + flags, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout)
    assert d["n_gpus"] == 8 and d["transport"] == "rccl"
    assert len(d["roofline"]["frac_by_rank"]) == 8 and all(f > 0 for f in d["roofline"]["frac_by_rank"])
    assert len(d["sustained"]["frac_last_window_by_rank"]) == 8
    assert len(d["numa_by_rank"]) == 8 and all("pinned" in n for n in d["numa_by_rank"])
    assert [b["rank"] for b in d["by_rank"]] == list(range(8))
    pg = d["peaks_allgather"]
    assert pg["peaks_by_rank"] == [1] * 8 and pg["values_expected"] == 8 == pg["values_gathered"] and pg["ok"] is True
    assert d["scan"]["scan_ms_max_over_ranks"] > 0 and d["peaks_allgather_us"] > 0 and d["rccl_proof"] is not None
    rt = d["realtime_per_gpu"]
    assert rt["front_ends_per_gpu"] == 6 and rt["ok_by_rank"] == [True] * 5 + [False] + [True] * 2
    assert rt["front_ends_sustained_total"] == 7 * 6 and rt["fm_channels_sustained_total"] == 7 * 6 * 32
    assert rt["deadline_misses_by_rank"][5] == 3 * 2 and rt["errors"] == []


def test_numa_pinning_reads_sysfs_and_survives_its_absence(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    before = os.sched_getaffinity(0)
    try:
        out = bench.pin_to_gpu_numa(0)                      # this container: no AMD render node, or one without NUMA info
        assert out["pinned"] in (True, False) and "numa_node" in out
        assert os.sched_getaffinity(0) <= before and len(os.sched_getaffinity(0)) >= 1
    finally:
        os.sched_setaffinity(0, before)
