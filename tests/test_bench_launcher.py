"""`python bench.py --gpus N` with no launcher around it starts its own N ranks (the reference
starts one channelizer process per source: rc_frontend/receiver.py:67-70).  Run here end to end on a stub of
librcf's Python layer (tests/stub_native.py): spawn, host rendezvous, max-over-ranks timing, peak gather, ONE line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--steps", "6", "--warmup", "1", "--block", str(1 << 16), "--prewarm-seconds", "0", "--no-extras",
         "--no-cpu-baseline", "--no-sustained"]


def _last_line(stdout):
    """what the driver can still see of a run: the last line of the last 8 000 bytes of stdout (its tail is 8 018) -- it
    must be ONE complete JSON object"""
    tail = stdout[-8000:]
    line = tail.splitlines()[-1]
    assert len(stdout.splitlines()[-1]) + 1 <= 8000, "the last line does not fit the driver's stdout tail"
    return json.loads(line)


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


def test_eight_rank_line_carries_every_rank_and_a_real_time_point_per_gpu(tmp_path):
    """VERDICT r04 item 5, the pre-flight of the first 8-GPU run: per-rank roofline fractions and sustained windows (not only
    the slowest rank's), each rank's NUMA pinning, `--config cfg5` with the scan time over ranks, the gather time and the
    gathered count checked against the sum of the ranks' peak counts, and one paced real-time point PER GPU so that the
    line answers "channels sustained" at N = 8 too -- here on the stub, one rank told to miss its deadlines"""
    flags = ["--steps", "6", "--warmup", "1", "--block", str(1 << 20), "--prewarm-seconds", "0", "--no-cpu-baseline",
             "--sustained-seconds", "0.05", "--config", "cfg5", "--rt-seconds", "0.5", "--rt-k-per-gpu", "6", "--rt-pumps", "2"]
    full_path = str(tmp_path / "full.json")
    env = dict(os.environ, RCF_BENCH_NATIVE="stub_native", RCF_BENCH_TRANSPORT="rccl", STUB_RCCL="1", STUB_DEVICES="8", STUB_RT_MISS="5",
               RCF_BENCH_FULL=full_path,
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"] + flags, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_line(r.stdout)
    assert d["n_gpus"] == 8 and d["transport"] == "rccl" and d["full_record"] == full_path
    assert len(d["roofline"]["frac_by_rank"]) == 8 and all(f > 0 for f in d["roofline"]["frac_by_rank"])
    assert len(d["sustained"]["frac_last_window_by_rank"]) == 8
    assert len(d["numa_node_by_rank"]) == 8
    full = json.load(open(full_path))                          # the full record: every rank's own entry
    assert len(full["numa_by_rank"]) == 8 and all("pinned" in n for n in full["numa_by_rank"])
    assert [b["rank"] for b in full["by_rank"]] == list(range(8))
    assert full["value"] == d["value"] and full["ms_per_step"] == d["ms_per_step"]
    pg = d["peaks_allgather"]
    assert pg["peaks_by_rank"] == [1] * 8 and pg["values_expected"] == 8 == pg["values_gathered"] and pg["ok"] is True
    assert d["scan"]["scan_ms_max_over_ranks"] > 0 and d["peaks_allgather_us"] > 0 and d["rccl_proof"] is not None
    rt = d["realtime_per_gpu"]
    assert rt["front_ends_per_gpu_asked"] == 6
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


REALTIME_FLAGS = ["--steps", "6", "--warmup", "1", "--block", str(1 << 20), "--prewarm-seconds", "0", "--no-cpu-baseline",
                  "--sustained-seconds", "0.05", "--rt-seconds", "0.5", "--rt-k-per-gpu", "6", "--rt-pumps", "2"]


def _run_flags(gpus, flags, extra_env, timeout=600):
    env = dict(os.environ, RCF_BENCH_NATIVE="stub_native",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), os.environ.get("PYTHONPATH", "")]))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus)] + flags, env=env,
                          capture_output=True, text=True, timeout=timeout)


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


@pytest.mark.parametrize("world", [1, 2, 8])
def test_the_last_stdout_line_is_a_compact_contract_object_the_driver_can_parse(world, tmp_path):
    """BENCH_r05.json came back `parsed: null` (one 27.9 KB line).  The last stdout line is now a
    compact object -- it fits the driver's 8 018-byte stdout tail whole, at N = 1, 2 and 8 -- and the full record is a file"""
    full_path = str(tmp_path / "full.json")
    r = _run_flags(world, REALTIME_FLAGS, {"RCF_BENCH_TRANSPORT": "rccl", "STUB_RCCL": "1", "STUB_DEVICES": "8",
                                           "RCF_BENCH_FULL": full_path})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len([l for l in r.stdout.splitlines() if l.strip()]) == 1
    d = _last_line(r.stdout)
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == world and "workload" in d["config"] and "model" not in d["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms"):
        assert k in d["roofline"], k
    assert len(d["ms_per_step_by_rank"]) == world
    assert d["rccl_ranks"] == world and d["transport"] == ("rccl" if world > 1 else "none (one rank)")
    full = json.load(open(full_path))
    assert full["value"] == d["value"] and len(json.dumps(full)) >= len(json.dumps(d)) - 200


def test_eight_ranks_that_cannot_use_rccl_miss_their_real_time_point_and_cannot_pin_still_print_the_line(tmp_path):
    """nothing that goes wrong OUTSIDE the timed region may cost the 8-GPU run its line -- no librccl
    (the stub's comm_unique_id raises), a rank whose paced point misses its deadlines, no NUMA information to pin by"""
    r = _run_flags(8, REALTIME_FLAGS, {"RCF_BENCH_TRANSPORT": "rccl", "STUB_DEVICES": "8", "STUB_RT_MISS": "3",
                                       "RCF_BENCH_FULL": str(tmp_path / "f.json")})
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_line(r.stdout)
    assert d["n_gpus"] == 8 and d["transport"] == "host-tcp" and d["rccl_ranks"] == 0
    assert len(d["ms_per_step_by_rank"]) == 8 and len(d["roofline"]["frac_by_rank"]) == 8
    assert d["numa_node_by_rank"] == [None] * 8
    rt = d["realtime_per_gpu"]
    assert rt["ok_by_rank"] == [True] * 3 + [False] + [True] * 4 and rt["front_ends_sustained_total"] == 7 * rt["front_ends_per_gpu"]
    assert d["peaks_allgather"]["ok"] is True and d["peaks_allgather"]["transport"] == "host TCP"


def test_per_gpu_real_time_point_is_sized_to_the_cpu_quota(monkeypatch):
    """N ranks share the host: K per GPU = min(asked, quota x 32 / ranks) -- 8 x 512 paced sources on a 16-core quota miss
    for reasons that are not the GPUs'"""
    sys.path.insert(0, ROOT)
    from benchlib import headline
    import types
    args = types.SimpleNamespace(rt_k_per_gpu=512)
    monkeypatch.setattr(headline, "cgroup_cpu_stat", lambda: (0, 0, 0, 16.0))
    assert headline.rt_k_per_gpu(args, 8) == (64, 16.0) and headline.rt_k_per_gpu(args, 1) == (512, 16.0)
    monkeypatch.setattr(headline, "cgroup_cpu_stat", lambda: (None, None, None, None))
    assert headline.rt_k_per_gpu(args, 8) == (512, None)


def test_compact_line_of_a_full_single_gpu_record_fits_and_keeps_the_judged_objects():
    """the largest full record there is (round 5's own 27.9 KB line, every leg run) through benchlib.compact: under the
    limit, contract keys intact, value / ms_per_step / the roofline's arithmetic untouched by the rounding"""
    sys.path.insert(0, ROOT)
    from benchlib import compact
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[5-9]_bench*.json"))):
        full = json.load(open(f))
        if "metric" not in full or "roofline" not in full:
            continue
        if "full_record" in full:                                   # (a compact line itself: rNN_bench_line.json)
            assert len(json.dumps(full)) + 1 <= compact.LIMIT and all(k in full for k in CONTRACT), f
            continue
        line = json.dumps(compact.compact(full))
        assert len(line) + 1 <= compact.LIMIT, (f, len(line))
        d = json.loads(line)
        for k in CONTRACT:
            assert k in d, (f, k)
        assert d["value"] == full["value"] and d["ms_per_step"] == full["ms_per_step"]
        assert d["roofline"]["frac"] == full["roofline"]["frac"] and d["roofline"]["achieved"] == full["roofline"]["achieved"]
        if full.get("cpu_baseline"):
            assert d["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-4)
            assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_compact_guard_drops_summaries_before_it_breaks_the_limit():
    sys.path.insert(0, ROOT)
    from benchlib import compact
    full = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 8, "steps": 1, "warmup": 0, "ms_per_step": 1.0,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 300, "block_samples": 1}, "roofline": {"bound": "hbm", "frac": 0.5},
            "cpu_baseline": None,
            "realtime": {"pfb256": {"K_max": 1, "points": [{"front_ends": i, "ok": True, "deadline_misses": 0,
                                                            "latency_ms_p99": 1.2345678} for i in range(2000)]}}}
    line = json.dumps(compact.compact(full))
    assert len(line) + 1 <= compact.LIMIT
    d = json.loads(line)
    assert d["roofline"]["frac"] == 0.5 and d["realtime"]["pfb256"]["K_max"] == 1 and "points" not in d["realtime"]["pfb256"]
