#!/usr/bin/env python3
"""bench.py -- headline benchmark: BASELINE.json configs[1]

  256-channel polyphase filterbank over one 20 Msps synthetic IQ stream per MI355X, stage-2 xlating
  FIR (/3) + FM discriminator on 32 active bins.

One "step" = one commit of a BLOCK-sample batch that is already resident in HBM: PFB kernel (all 256 bins
written), 32 stage-2 FIRs with their discriminators fused in, history carry-over.  value = input IQ Msamples/s
over all ranks.  N > 1: one independent 20 Msps front-end per GPU (config_denver_massive_p25-style, BASELINE
configs[3]); weak scaling; no data-path collective -- the only RCCL traffic is the all-gather of detected-peak
lists (ncclAllGather through librcf's C ABI) after the timed region, reported as peaks_allgather_us.

OUTPUT.  The LAST (and only) stdout line is a compact JSON object of <= 7 600 bytes (benchlib/compact.py): the
contract keys, `roofline`, `cpu_baseline` and one-number summaries of every leg.  The full record -- every point of
every leg -- is written to bench_full.json (cwd, and gpurun_out/ which is what comes back from a GPU box); the
compact line names it in `full_record`.

No PyTorch: device work goes through librcf's C ABI (ctypes), ranks meet over rcf.multigpu.HostGroup (TCP next
to MASTER_PORT), the barrier on both sides of the timed region is `rcf_allreduce_max` (stream sync + one
ncclAllReduce), which also yields the max-over-ranks time.  The process pins itself to the CPUs of its GPU's NUMA
node before the HIP runtime starts (N = 1 too).

The legs (benchlib/, one module each; all but the first outside the timed region, rank 0 at N = 1):
  headline.py   the timed configuration, its roofline (HIP events attached to the filterbank's dispatch), the
                filterbank alone, `sustained` (>= 2 s of commits, launch time per window of 100, shader clock)
  traffic.py    roofline.traffic: two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) of a short headline run
  banks.py      channels.direct_bank: the reference-shaped bank (one 2909-tap xlating FIR /800 + discriminator per
                channel) opened and run at 256 .. 196608 channels;  channels.reference_grid_filterbank: the 1600-bin
                bank whose bins are the reference's channels (+ taps, + the discriminator fused into the bank)
  scan.py       scan: BASELINE configs[2] (1M-point FFT x 1000 frames / 100-frame average + peak pick);  scan_ref: the
                reference's own size (fs = 2.4 Msps, N = 16384, 1000 / 100), each with its roofline
  daemon.py     daemon (the product's own data plane: receiver + native pump + egress thread, 32 x 20 Msps x 64 channels)
  ingest.py     end_to_end (PCIe-inclusive ingest: never `value`), control_plane (100 x create / release)
  group.py      group_capacity: one grouped launch per stage over 80 front-ends, resident
  realtime.py   the paced leg: K independent 20 Msps u8 front-ends fed at WALL-CLOCK rate by native pump threads
                (rcf_pump_*), K raised until a block misses its deadline -> K_max, confirmed over --rt-seconds
  cpu.py        cpu_baseline: the oracle's C port of the reference path on the host cores (threads = what the
                container's CPU quota allows, throttle counters reported) + the parity check of the timed
                configuration's FM outputs against the oracle

--config cfg5 = BASELINE configs[4]'s per-GPU shape instead (25 Msps slice, 512-bin bank, N = 2^20 / 1000 / 100 scan
on the slice, <= 1024 peaks per rank into the all-gather); the default (cfg4) is configs[1] / configs[3].

Launch: python bench.py [--gpus N --steps K --warmup W].  For N > 1 either a launcher starts the ranks
        (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...: RANK / LOCAL_RANK /
        WORLD_SIZE / MASTER_* come from the environment), or -- when WORLD_SIZE is not set -- bench.py starts its own
        N rank processes (one per GPU, the way the reference starts one channelizer process per source:
        rc_frontend/receiver.py:67-70, systemd/radiocapture-channelizer@.service:11), waits for them and relays rank
        0's line.  Fewer than N visible devices is an error (rc 2), not a silent downgrade; RCF_BENCH_DEVICE=<d> puts
        every rank on device d (the N > 1 code on a 1-GPU box: host transport, RCCL cannot span one device twice).
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "radiocapture-rf_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from benchlib.common import FS, NB, N_ACTIVE, HBM_PEAK_GBS, proto_taps, read_sclk_mhz, cgroup_cpu_stat  # noqa: E402,F401
from benchlib.launcher import pin_to_gpu_numa, load_native, spawn_ranks  # noqa: E402,F401
from benchlib.sustained import sustained_leg  # noqa: E402,F401
from benchlib.banks import direct_bank_sweep, reference_grid_leg  # noqa: E402,F401
from benchlib.scan import scan_leg, scan_ref_leg  # noqa: E402,F401
from benchlib.ingest import end_to_end_leg, control_plane_leg  # noqa: E402,F401
from benchlib.group import group_capacity_leg  # noqa: E402,F401
from benchlib.realtime import realtime_point, realtime_leg  # noqa: E402,F401
from benchlib.daemon import daemon_leg  # noqa: E402,F401
from benchlib.cpu import cpu_baseline  # noqa: E402,F401
from benchlib import compact, headline  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--block", type=int, default=1 << 25, help="samples per step (resident batch)")
    ap.add_argument("--config", choices=["cfg4", "cfg5"], default="cfg4",
                    help="cfg4: BASELINE configs[1] per GPU (x N = configs[3]); cfg5: BASELINE configs[4]'s per-GPU shape")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed legs (sweep, scan, end-to-end, ...)")
    ap.add_argument("--no-sustained", action="store_true", help="skip the >= 2 s sustained leg")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="length of the sustained leg")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run")
    ap.add_argument("--alone-warm", type=int, default=300,
                    help="untimed commits before the filterbank-alone pass (the chip's ramp after an idle queue is ~15 ms)")
    ap.add_argument("--alone-launches", type=int, default=100, help="timed launches of the filterbank-alone pass")
    ap.add_argument("--time-every", type=int, default=4,
                    help="HIP events around every n-th filterbank launch of the timed region (1 = all of them)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed commits before the warm-up steps: the metric is SUSTAINED throughput, and the chip needs "
                         "~100 launches (15 ms) after idling before the filterbank launch settles; 0 = none")
    ap.add_argument("--sweep-max", type=int, default=196608, help="largest direct-bank channel count to open")
    ap.add_argument("--rt-seconds", type=float, default=10.0, help="seconds of the paced leg's confirmation run (0 = skip the leg)")
    ap.add_argument("--rt-block-ms", type=float, default=20.0, help="block length of the paced real-time leg")
    ap.add_argument("--rt-k-first", type=int, default=512, help="front-end count the real-time search starts at")
    ap.add_argument("--rt-k-cap", type=int, default=1280, help="largest front-end count the real-time search tries")
    ap.add_argument("--rt-k-per-gpu", type=int, default=512,
                    help="N > 1: front-ends of the one paced point every rank runs (capped by CPU quota x 32 / ranks)")
    ap.add_argument("--rt-pumps", type=int, default=0, help="native pump threads (groups) of the real-time leg (0: four)")
    ap.add_argument("--rt-window-ms", type=float, default=1.0, help="batching window of the pumps")
    ap.add_argument("--rt-shapes", default="pfb256,grid1600,grid1600fm", help="shapes of the real-time leg")
    ap.add_argument("--rt-burst", action="store_true",
                    help="real-time leg: every front-end's block completes at the same instant (default: spread over the period)")
    ap.add_argument("--full-on-stdout", action="store_true", help="print the FULL record as the one line (pre-round-6 behaviour)")
    return ap.parse_args(argv)


def leg(fn, *a, **k):
    """a leg outside the timed region must not cost the run its line"""
    try:
        return fn(*a, **k)
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def untimed_legs(args, out, ctx, native, synth, device):
    """rank 0, N = 1, cfg4: the bandwidth-bound legs first -- seconds of matrix-core work at full power (the sweep) leave
    the chip at lower clocks for whatever runs next (the same filterbank launch 0.112 ms before it, 0.139 after)"""
    tile, carriers = ctx["tile"], ctx["meta"]["carriers"]
    out["channels"]["reference_grid_filterbank"] = leg(reference_grid_leg, native, tile, device)
    out["scan"] = leg(scan_leg, native, synth, device)
    out["scan_ref"] = leg(scan_ref_leg, native, synth, device)
    out["end_to_end"] = leg(end_to_end_leg, native, tile, device)
    out["group_capacity"] = leg(group_capacity_leg, native, tile, carriers, device)
    if args.rt_seconds > 0:
        out["realtime"] = leg(realtime_leg, native, tile, carriers, device, seconds=args.rt_seconds, block_ms=args.rt_block_ms,
                              k_first=args.rt_k_first, k_cap=args.rt_k_cap, stagger=not args.rt_burst, n_pumps=args.rt_pumps,
                              window_ms=args.rt_window_ms, shapes=tuple(x for x in args.rt_shapes.split(",") if x))
    if args.rt_seconds > 0:
        out["daemon"] = leg(daemon_leg, device)
    out["control_plane"] = leg(control_plane_leg, device)
    counts = [c for c in (256, 1024, 4096, 16384, 65536, 131072, 196608) if c <= args.sweep_max]
    out["channels"]["direct_bank"] = leg(direct_bank_sweep, native, tile, device, counts)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    R = headline.Ranks()
    if args.gpus != R.n_gpus and R.rank == 0:
        print("note: --gpus %d but WORLD_SIZE=%d; using %d" % (args.gpus, R.world, R.n_gpus), file=sys.stderr)
    # this rank's threads stay on the CPUs of its GPU's NUMA node (before the HIP runtime starts its own threads, and
    # before any pinned host memory is allocated): N = 1 too -- the paced leg's pumps and rings sit next to the GPU's root
    numa = pin_to_gpu_numa(R.local_rank) if os.environ.get("RCF_BENCH_NO_PIN", "0") in ("", "0") else None
    # (the host driver only supports dmabuf IPC: without this RCCL's peer mappings fail -- set before the HIP runtime loads)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from rcf import multigpu, synth
    native = load_native()
    if native.device_count() < 1:
        raise RuntimeError("bench.py needs an MI355X (no HIP device visible)")
    if native.device_count() <= R.local_rank:
        raise RuntimeError("rank %d: device %d asked for, %d visible" % (R.rank, R.local_rank, native.device_count()))

    out, ctx = headline.run(args, R, native, multigpu, synth, numa)
    if R.rank != 0:
        return
    if R.n_gpus == 1 and not args.no_extras and not ctx["cfg5"]:
        untimed_legs(args, out, ctx, native, synth, R.local_rank)
    if R.n_gpus == 1 and not args.no_cpu_baseline:
        if ctx["cfg5"]:
            # same structure on the slice's stream: the reference would run one 3637-tap xlating FIR /1000 per channel
            out["cpu_baseline"] = leg(cpu_baseline, ctx["tile"][: 1 << 20], ctx["meta"]["carriers"], None, FS=ctx["fs"])
        else:
            out["cpu_baseline"] = leg(cpu_baseline, ctx["tile"], ctx["meta"]["carriers"], ctx["fm_check"])
        db = out["channels"].get("direct_bank")
        cb = out["cpu_baseline"]
        if db and "channels_run_in_real_time" in db and "largest_cpu_realtime_channels" in cb:
            # against the LARGEST of the CPU figures (measured reference structure, time-tiled best CPU, SURVEY's formula
            # over every physical core of the box)
            cb["gpu_channels_run_in_real_time_over_cpu_realtime_channels"] = (
                db["channels_run_in_real_time"] / cb["largest_cpu_realtime_channels"])
            cb["gpu_over_cpu_note"] = ("%d reference-shaped channels opened and run in real time on one MI355X (direct "
                                       "bank) / %.0f, the largest CPU figure (%s)" % (
                                           db["channels_run_in_real_time"], cb["largest_cpu_realtime_channels"], cb["largest_is"]))
    else:
        out["cpu_baseline"] = None
    if args.full_on_stdout:
        print(json.dumps(out))
        return
    line = json.dumps(compact.compact(out, compact.write_full(out)))
    assert len(line) + 1 <= 8000, "the compact line grew to %d bytes" % len(line)
    sys.stdout.flush()
    print(line, flush=True)


if __name__ == "__main__":
    main()
